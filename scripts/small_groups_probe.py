#!/usr/bin/env python3
"""Sixteen / eight recordings of 1 h (5 400 x 256) through fa_ahc_linkage_batch as one uniform batch or as two / three batches side by side."""
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

ctx = fa.default_context()
probs = []
for k in range(16):
    x = e2e_session(1.0, 12, seed=50 + k)["emb"].astype(np.float64)
    probs.append(x / np.sqrt((x * x).sum(axis=1, keepdims=True)))
ref = [fa.linkage(x, ctx=ctx)[1] for x in probs]
out = []
for K in (8, 16):
    for g in ("1", "2", "3", None):
        os.environ.pop("FA_AHC_UNI_GROUPS", None)
        if g:
            os.environ["FA_AHC_UNI_GROUPS"] = g
        fa.linkage_batch(probs[:K], ctx=ctx)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            st, zs = fa.linkage_batch(probs[:K], ctx=ctx)
            best = min(best, time.perf_counter() - t0)
        same = all(s == 0 and np.array_equal(z, r) for s, z, r in zip(st, zs, ref))
        rec = {"n": 5400, "K": K, "groups": g or "default", "wall_s_best_of_3": round(best, 4), "audio_hours_per_s": K / best, "equal_single": bool(same)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
with open(os.path.join(ROOT, "gpurun_out", "summary", "small_groups_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
