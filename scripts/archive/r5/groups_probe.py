#!/usr/bin/env python3
"""Uniform batches side by side: 2 / 3 groups at two slots per thread, K = 8 and 12 recordings of 8 h (linkage only)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa
from e2e_inputs import e2e_session
ctx = fa.default_context()
def unit_rows(hours, seed):
    x = e2e_session(hours, 12, seed=seed)["emb"].astype(np.float64)
    return x / np.sqrt((x * x).sum(axis=1, keepdims=True))
big = [unit_rows(8.0, 5 + k) for k in range(12)]
for K in (8, 12):
    for groups in ("2", "3"):
        for cpt in ("2", "1"):
            os.environ["FA_AHC_UNI_GROUPS"] = groups; os.environ["FA_AHC_UNI_CPT"] = cpt
            ctx.trim()
            fa.linkage_batch(big[:K], ctx=ctx)
            t0 = time.perf_counter()
            st, zs, stats = fa.linkage_batch(big[:K], ctx=ctx, return_stats=True)
            wall = time.perf_counter() - t0
            print(json.dumps({"K": K, "groups": groups, "cpt": cpt, "wall_s": round(wall, 4), "init_ms": stats[0]["init_ms"], "merge_ms": stats[0]["merge_ms"],
                              "us_per_round": 1e3 * stats[0]["merge_ms"] / stats[0]["rounds"], "audio_hours_per_s_linkage_only": K * 8 / wall, "ok": st == [0] * K}), flush=True)
