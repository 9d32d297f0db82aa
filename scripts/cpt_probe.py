#!/usr/bin/env python3
"""Round 5: slots per thread of the AHC round kernel (ahc_round_body's CPT).  (1) every form against the reference build (oracle/_ref) on inputs the
CPU finishes in seconds, single problems (FA_AHC_CPT) and uniform batches (FA_AHC_UNI_CPT); (2) round time and throughput of uniform batches of 8 h
(43 200 x 256) and 1 h (5 400 x 256) recordings per CPT and group count, and of the single chain per CPT."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa  # noqa: E402
import oracle  # noqa: E402  (the checker of this probe, not the thing measured)
from e2e_inputs import e2e_session  # noqa: E402

ctx = fa.default_context()
quick = "--quick" in sys.argv


def setenv(env):
    for k in ("FA_AHC_CPT", "FA_AHC_UNI_CPT", "FA_AHC_UNI_GROUPS", "FA_AHC_NO_UNIFORM", "FA_AHC_NO_SINGLE_BLOCK"):
        os.environ.pop(k, None)
    os.environ.update(env)


def unit_rows(hours, seed):
    x = e2e_session(hours, 12, seed=seed)["emb"].astype(np.float64)
    return x / np.sqrt((x * x).sum(axis=1, keepdims=True))


# ---- (1) parity with the reference build
rng = np.random.default_rng(0)
bad = 0
cases = []
for n, d in ((300, 16), (513, 32), (700, 64), (1024, 8), (1500, 24), (2500, 32)):
    x = oracle.ahc_normalize(rng.standard_normal((n, d)))
    cases.append((x, oracle.linkage_ref(x)[1]))
dup = oracle.ahc_normalize(rng.standard_normal((400, 8)))
dup = np.concatenate([dup, dup[:150]])            # exact ties -> the reference-order route
cases.append((dup, oracle.linkage_ref(dup)[1]))
for cpt in ("1", "2", "4"):
    for mode in (fa.AHC_MODE_AUTO, fa.AHC_MODE_EXACT):
        for extra in ({}, {"FA_AHC_NO_SINGLE_BLOCK": "1"}):
            setenv({"FA_AHC_CPT": cpt, **extra})
            for x, zr in cases:
                if mode == fa.AHC_MODE_EXACT and x is dup:
                    continue                      # EXACT keeps its own documented tie order
                st, z = fa.linkage(x, ctx=ctx, mode=mode)
                ok = st == 0 and np.array_equal(z, zr)
                bad += not ok
                if not ok:
                    print(json.dumps({"FAIL": "single", "cpt": cpt, "mode": mode, "n": len(x), "extra": extra, "status": st}), flush=True)
for cpt in ("1", "2", "4"):
    setenv({"FA_AHC_UNI_CPT": cpt})
    base = cases[5][0]                                                              # 2 500 x 32
    group = [base, base[:2100], base[:1800], base[:2300], base[:1300]]              # sizes within a factor of two: one uniform batch
    refs = [oracle.linkage_ref(x)[1] for x in group]
    st, zs = fa.linkage_batch(group, ctx=ctx)
    for i, (s, z) in enumerate(zip(st, zs)):
        ok = s == 0 and np.array_equal(z, refs[i])
        bad += not ok
        if not ok:
            print(json.dumps({"FAIL": "uniform", "cpt": cpt, "i": i, "n": len(group[i]), "status": s}), flush=True)
print(json.dumps({"parity_failures": bad}), flush=True)

# ---- (2) timing
out = []
setenv({})
K8 = 4 if quick else 12
big = [unit_rows(8.0, 5 + k) for k in range(K8)]
setenv({"FA_AHC_CPT": "1"})
ref = [fa.linkage(x, ctx=ctx)[1] for x in big]
for cpt in ("1", "2", "4"):
    setenv({"FA_AHC_CPT": cpt})
    fa.linkage(big[0], ctx=ctx)
    t0 = time.perf_counter()
    st, z, stats = fa.linkage(big[0], ctx=ctx, return_stats=True)
    wall = time.perf_counter() - t0
    rec = {"what": "single chain", "n": 43200, "cpt": cpt, "wall_s": round(wall, 4), "us_per_round": 1e3 * stats["merge_ms"] / max(1, stats["rounds"]), "init_ms": stats["init_ms"],
           "equal": bool(st == 0 and np.array_equal(z, ref[0]))}
    print(json.dumps(rec), flush=True)
    out.append(rec)
ctx.trim()
for K in ((2, 4) if quick else (2, 4, 8, 12)):
    for cpt in ("1", "2", "4"):
        for groups in ("1", "2"):
            if groups == "2" and K < 4:
                continue
            setenv({"FA_AHC_UNI_CPT": cpt, "FA_AHC_UNI_GROUPS": groups})
            ctx.trim()
            fa.linkage_batch(big[:K], ctx=ctx)
            t0 = time.perf_counter()
            st, zs, stats = fa.linkage_batch(big[:K], ctx=ctx, return_stats=True)
            wall = time.perf_counter() - t0
            same = all(s == 0 and np.array_equal(z, ref[i]) for i, (s, z) in enumerate(zip(st, zs)))
            rec = {"what": "uniform batch", "n": 43200, "K": K, "cpt": cpt, "groups": groups, "wall_s": round(wall, 4), "merge_ms": stats[0]["merge_ms"], "init_ms": stats[0]["init_ms"],
                   "us_per_round": 1e3 * stats[0]["merge_ms"] / max(1, stats[0]["rounds"]), "audio_hours_per_s_linkage_only": K * 8.0 / wall, "equal_single": bool(same)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
ctx.trim()
small = [unit_rows(1.0, 50 + k) for k in range(16)]
setenv({})
sref = [fa.linkage(x, ctx=ctx)[1] for x in small]
for K in (4, 16):
    for cpt in ("1", "2", "4"):
        setenv({"FA_AHC_UNI_CPT": cpt, "FA_AHC_UNI_GROUPS": "1"})
        fa.linkage_batch(small[:K], ctx=ctx)
        t0 = time.perf_counter()
        st, zs, stats = fa.linkage_batch(small[:K], ctx=ctx, return_stats=True)
        wall = time.perf_counter() - t0
        same = all(s == 0 and np.array_equal(z, sref[i]) for i, (s, z) in enumerate(zip(st, zs)))
        rec = {"what": "uniform batch", "n": 5400, "K": K, "cpt": cpt, "wall_s": round(wall, 4), "merge_ms": stats[0]["merge_ms"], "init_ms": stats[0]["init_ms"],
               "us_per_round": 1e3 * stats[0]["merge_ms"] / max(1, stats[0]["rounds"]), "audio_hours_per_s_linkage_only": K * 1.0 / wall, "equal_single": bool(same)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
# short recordings: the single-block form now reaches 1 024 points
for n in (200, 400, 900, 1024, 2000):
    x = oracle.ahc_normalize(np.random.default_rng(n).standard_normal((n, 256)))
    for env in ({}, {"FA_AHC_NO_SINGLE_BLOCK": "1"}):
        setenv(env)
        fa.linkage(x, ctx=ctx)
        t0 = time.perf_counter()
        for _ in range(5):
            st, z = fa.linkage(x, ctx=ctx)
        rec = {"what": "short recording", "n": n, "env": env, "ms_per_call": 1e3 * (time.perf_counter() - t0) / 5}
        print(json.dumps(rec), flush=True)
        out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r5", "cpt_probe.json"), "w") as f:
    json.dump({"parity_failures": bad, "records": out}, f, indent=1)
