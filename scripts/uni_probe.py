#!/usr/bin/env python3
"""Round time of the uniform-layout batched AHC round (ahc_round_uni) against the number of problems per launch and the kernel's register
budget (FA_AHC_UNI_WAVES = waves per SIMD the build allows: default 5 = 94 VGPRs, 6 = 80 + 52 B scratch, 8 = 64 + 120 B scratch), next to the round-2 batched
kernel (FA_AHC_NO_UNIFORM) and the round-3 chains in flight (FA_AHC_IN_FLIGHT).  Problems: the 8 h session (43 200 x 256) and 1 h (5 400 x 256)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import fluidaudio_amd as fa  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

ctx = fa.default_context()


def unit_rows(hours, seed):
    x = e2e_session(hours, 12, seed=seed)["emb"].astype(np.float64)
    return x / np.sqrt((x * x).sum(axis=1, keepdims=True))


_last_shape = [None]


def run(probs, env):
    for k in ("FA_AHC_UNI_WAVES", "FA_AHC_UNI_GROUPS", "FA_AHC_NO_UNIFORM", "FA_AHC_IN_FLIGHT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    shape = (len(probs), env.get("FA_AHC_UNI_GROUPS"), env.get("FA_AHC_IN_FLIGHT"))
    if shape != _last_shape[0]:
        ctx.trim()                      # the workspaces of another split must not add up to more than HBM holds
        _last_shape[0] = shape
    fa.linkage_batch(probs, ctx=ctx)
    t0 = time.perf_counter()
    st, zs, stats = fa.linkage_batch(probs, ctx=ctx, return_stats=True)
    wall = time.perf_counter() - t0
    return st, zs, stats, wall


out = []
big = [unit_rows(8.0, 5 + k) for k in range(12)]
ref = {}
for k in range(12):
    st, z = fa.linkage(big[k], ctx=ctx)
    assert st == 0
    ref[k] = z
for K in (1, 2, 4, 6, 8, 12):
    for env in ({"FA_AHC_UNI_GROUPS": "1"}, {"FA_AHC_UNI_GROUPS": "2"}, {"FA_AHC_UNI_GROUPS": "3"}, {"FA_AHC_UNI_GROUPS": "4"}, {}, {"FA_AHC_NO_UNIFORM": "1"}, {"FA_AHC_IN_FLIGHT": "1"}):
        if env.get("FA_AHC_IN_FLIGHT") and K > 4:
            continue
        if env.get("FA_AHC_NO_UNIFORM") and K not in (2, 4, 8):
            continue
        if K == 1 and env:
            continue
        g = int(env.get("FA_AHC_UNI_GROUPS", "1"))
        if g > 1 and K < 2 * g:
            continue
        st, zs, stats, wall = run(big[:K], env)
        same = all(s == 0 and np.array_equal(z, ref[i]) for i, (s, z) in enumerate(zip(st, zs)))
        rec = {"n": 43200, "K": K, "env": env, "wall_s": round(wall, 4), "merge_ms": stats[0]["merge_ms"], "init_ms": stats[0]["init_ms"], "rounds": stats[0]["rounds"],
               "us_per_round": 1e3 * stats[0]["merge_ms"] / max(1, stats[0]["rounds"]), "audio_hours_per_s_linkage_only": K * 8.0 / wall, "equal_single": bool(same)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    ctx.trim()
small = [unit_rows(1.0, 50 + k) for k in range(16)]
sref = [fa.linkage(x, ctx=ctx)[1] for x in small]
for K in (4, 16):
    for env in ({}, {"FA_AHC_NO_UNIFORM": "1"}):
        st, zs, stats, wall = run(small[:K], env)
        same = all(s == 0 and np.array_equal(z, sref[i]) for i, (s, z) in enumerate(zip(st, zs)))
        rec = {"n": 5400, "K": K, "env": env, "wall_s": round(wall, 4), "merge_ms": stats[0]["merge_ms"], "init_ms": stats[0]["init_ms"], "rounds": stats[0]["rounds"],
               "us_per_round": 1e3 * stats[0]["merge_ms"] / max(1, stats[0]["rounds"]), "equal_single": bool(same)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
with open(os.path.join(ROOT, "gpurun_out", "summary", "uni_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
