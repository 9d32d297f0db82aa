#!/usr/bin/env python3
"""Round 5: what an input with exact ties costs.  43 200 x 256 (the 8 h session) with 30 % of its rows duplicated: AUTO halts at the first tied minimum and
the problem is recomputed in the reference's selection order (ahc_reforder.h); next to it the tie-free session and the reference-order mode on its own."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa
from e2e_inputs import e2e_session
ctx = fa.default_context()
x = e2e_session(8.0, 12, seed=5)["emb"].astype(np.float64)
x /= np.sqrt((x * x).sum(axis=1, keepdims=True))
n = len(x)
rng = np.random.default_rng(1)
dup = x.copy()
idx = rng.integers(0, n, int(0.3 * n)); src = rng.integers(0, n, int(0.3 * n))
dup[idx] = dup[src]
for name, data, mode in (("tie-free, AUTO", x, fa.AHC_MODE_AUTO), ("30 % duplicates, AUTO", dup, fa.AHC_MODE_AUTO), ("tie-free, REFERENCE_ORDER", x, fa.AHC_MODE_REFERENCE_ORDER),
                         ("30 % duplicates, REFERENCE_ORDER", dup, fa.AHC_MODE_REFERENCE_ORDER)):
    fa.linkage(data, mode=mode, ctx=ctx)
    t0 = time.perf_counter()
    st, z, stats = fa.linkage(data, mode=mode, ctx=ctx, return_stats=True)
    dt = time.perf_counter() - t0
    print(json.dumps({"case": name, "status": st, "wall_s": round(dt, 4), "reference_order": stats["reference_order"], "rounds": stats["rounds"], "merge_ms": stats["merge_ms"],
                      "us_per_row": 1e3 * stats["merge_ms"] / (n - 1)}), flush=True)
