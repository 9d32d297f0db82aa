#!/usr/bin/env python3
"""Round time of ONE merge chain (8 h session, 43 200 x 256; 50 000 x 256 iid) with the two builds of the single-problem round kernel:
FA_AHC_ROUND_BIG=1 forces the build that carries the many-record reduction (168 VGPRs), the default for N <= 65 536 is the 94-VGPR build."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa  # noqa: E402
from ahc_full_inputs import ahc_input  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

ctx = fa.default_context()
x = e2e_session(8.0, 12, seed=5)["emb"].astype(np.float64)
x /= np.sqrt((x * x).sum(axis=1, keepdims=True))
probs = {"8h session 43200x256": x, "iid 50000x256": ahc_input("iid", 50000, 256)}
out = []
for name, p in probs.items():
    ref = None
    for env in ({}, {"FA_AHC_ROUND_BIG": "1"}, {}, {"FA_AHC_ROUND_BIG": "1"}):
        os.environ.pop("FA_AHC_ROUND_BIG", None)
        os.environ.update(env)
        ctx.trim()                      # the cached graph is keyed by shape, not by kernel build: start from a fresh capture
        fa.linkage(p, ctx=ctx)
        st, z, s = fa.linkage(p, ctx=ctx, return_stats=True)
        ref = z if ref is None else ref
        rec = {"problem": name, "env": env, "us_per_round": 1e3 * s["merge_ms"] / max(1, s["rounds"]), "merge_ms": s["merge_ms"], "init_ms": s["init_ms"],
               "rounds": s["rounds"], "same_dendrogram": bool(np.array_equal(z, ref))}
        print(json.dumps(rec), flush=True)
        out.append(rec)
with open(os.path.join(ROOT, "gpurun_out", "summary", "single_chain_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
