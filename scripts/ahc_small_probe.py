#!/usr/bin/env python3
"""Latency of ONE linkage call through the drop-in symbol at small N (short recordings), next to the reference build on one host core
(run where oracle/_ref exists; test infrastructure)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
import fluidaudio_amd as fa  # noqa: E402
from conftest import speaker_mixture  # noqa: E402

try:
    import oracle
except Exception:  # noqa: BLE001
    oracle = None
out = []
for n in (50, 200, 500, 900, 2000, 4000):
    x = speaker_mixture(n, 256, 6, 0.04, n)
    fa.fastcluster_compute_centroid_linkage(x)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        st, z = fa.fastcluster_compute_centroid_linkage(x)
        ts.append(time.perf_counter() - t0)
    ref = None
    if oracle is not None:
        t0 = time.perf_counter()
        sr, zr = oracle.linkage_ref(x)
        ref = time.perf_counter() - t0
        assert np.array_equal(z, zr)
    out.append({"n": n, "gpu_ms": round(1e3 * min(ts), 3), "reference_cpu_ms": None if ref is None else round(1e3 * ref, 3)})
    print(out[-1])
print(json.dumps({"ahc_small": out}))
