#!/usr/bin/env python3
"""Latency of the whole clustering stage (fa_offline_cluster: AHC + VBx + centroids + constrained assignment) for SHORT recordings, next to
the CPU side (reference linkage build + C restatements, one core) — test infrastructure."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

try:
    import oracle
except Exception:  # noqa: BLE001
    oracle = None
ctx = fa.default_context()
out = []
for minutes in (1, 5, 20, 60):
    s = e2e_session(minutes / 60.0, 4 if minutes < 20 else 8)
    fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=ctx)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        res = fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=ctx)
        ts.append(time.perf_counter() - t0)
    cpu = None
    if oracle is not None:
        t0 = time.perf_counter()
        ref = oracle.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"])
        cpu = time.perf_counter() - t0
        assert np.array_equal(np.asarray(res.assignments), np.asarray(ref["assignments"]))
    out.append({"minutes": minutes, "embeddings": int(len(s["emb"])), "gpu_ms": round(1e3 * float(np.median(ts)), 3), "stages_s": {k: round(float(v), 6) for k, v in res.info.items() if k.endswith("_s")},
                "cpu_1_core_ms": None if cpu is None else round(1e3 * cpu, 3)})
    print(out[-1])
print(json.dumps({"cluster_small": out}))
