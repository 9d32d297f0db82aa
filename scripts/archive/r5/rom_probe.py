#!/usr/bin/env python3
"""Round 5: the reference-order run through the matrix filter (rom_scan / rom_select) against the matrix-free form (FA_AHC_RO_NO_MATRIX) on the 8 h session
(43 200 x 256), tie-free and with 30 % of its rows duplicated: dendrograms equal row for row, start-up / per-row times, and what a tie costs AUTO now."""
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
small = int(os.environ.get("ROM_PROBE_N", "0"))
if small:
    x, dup, n = x[:small], dup[:small], small
out = {}
def run(name, data, mode, env=None, reps=1):
    if env: os.environ[env] = "1"
    try:
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            st, z, stats = fa.linkage(data, mode=mode, ctx=ctx, return_stats=True)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]: best = (dt, st, z, stats)
    finally:
        if env: os.environ.pop(env, None)
    dt, st, z, stats = best
    print(json.dumps({"case": name, "status": st, "wall_s": round(dt, 4), "reference_order": stats["reference_order"], "rounds": stats["rounds"], "exact_rows": stats["rescans"],
                      "init_ms": round(stats["init_ms"], 2), "merge_ms": round(stats["merge_ms"], 2), "us_per_row": round(1e3 * stats["merge_ms"] / (n - 1), 3)}), flush=True)
    return z
if os.environ.get("ROM_PROBE_ONLY"):          # one run of one form (kernel traces)
    run("tie-free, REFERENCE_ORDER (matrix filter)", x, fa.AHC_MODE_REFERENCE_ORDER)
    sys.exit(0)
z_auto = run("tie-free, AUTO", x, fa.AHC_MODE_AUTO, reps=2)
z_rom = run("tie-free, REFERENCE_ORDER (matrix filter)", x, fa.AHC_MODE_REFERENCE_ORDER, reps=2)
z_mf = run("tie-free, REFERENCE_ORDER (matrix-free)", x, fa.AHC_MODE_REFERENCE_ORDER, env="FA_AHC_RO_NO_MATRIX")
print(json.dumps({"tie-free: matrix filter == matrix-free": bool(np.array_equal(z_rom, z_mf)), "== AUTO": bool(np.array_equal(z_rom, z_auto))}), flush=True)
run("tie-free, REFERENCE_ORDER (matrix filter, start-up by all N^2 / 2 exact sums)", x, fa.AHC_MODE_REFERENCE_ORDER, env="FA_AHC_ROM_DIRECT_START")
d_rom = run("30 % duplicates, REFERENCE_ORDER (matrix filter)", dup, fa.AHC_MODE_REFERENCE_ORDER, reps=2)
d_mf = run("30 % duplicates, REFERENCE_ORDER (matrix-free)", dup, fa.AHC_MODE_REFERENCE_ORDER, env="FA_AHC_RO_NO_MATRIX")
d_auto = run("30 % duplicates, AUTO (-> tie -> matrix filter)", dup, fa.AHC_MODE_AUTO, reps=2)
print(json.dumps({"duplicates: matrix filter == matrix-free": bool(np.array_equal(d_rom, d_mf)), "== AUTO": bool(np.array_equal(d_rom, d_auto))}), flush=True)
bad = np.nonzero((d_rom != d_mf).any(axis=1))[0]
if bad.size: print("first differing rows", bad[:5], d_rom[bad[0]], d_mf[bad[0]])
