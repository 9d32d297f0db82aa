#!/usr/bin/env python3
"""fa_ahc_linkage_batch vs sequential fa_ahc_linkage calls: K recordings of n x 256 speaker-mixture embeddings."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fluidaudio_amd as fa  # noqa: E402
from conftest import speaker_mixture  # noqa: E402

ctx = fa.default_context()
out = []
for K, n in ((16, 5400), (64, 1350), (4, 21600), (32, 5400)):
    probs = [speaker_mixture(n, 256, 8, 0.03, 100 + k) for k in range(K)]
    fa.linkage_batch(probs[:2], ctx=ctx)
    t0 = time.perf_counter(); st, zs, stats = fa.linkage_batch(probs, ctx=ctx, return_stats=True); tb = time.perf_counter() - t0
    assert all(s == 0 for s in st)
    fa.linkage(probs[0], ctx=ctx)
    t0 = time.perf_counter()
    seq = [fa.linkage(p, ctx=ctx) for p in probs]
    ts = time.perf_counter() - t0
    same = all(np.array_equal(z, s[1]) for z, s in zip(zs, seq))
    out.append({"recordings": K, "embeddings_each": n, "batch_s": tb, "batch_device_ms": stats[0]["total_ms"], "sequential_s": ts, "speedup": ts / tb,
                "identical_to_sequential": bool(same), "rounds_max": max(s["rounds"] for s in stats),
                "audio_hours_clustered_per_s": K * (n / 3 * 2 / 3600) / tb})
print(json.dumps({"ahc_batch": out, "note": "host-pointer entries (PCIe copies included); n embeddings = n/3 two-second windows"}))
