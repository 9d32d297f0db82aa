import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fluidaudio_amd as fa
from conftest import speaker_mixture
ctx = fa.default_context()
for n, dup in ((6000, 0.3), (20000, 0.3), (20000, 0.9)):
    x = speaker_mixture(n, 64, 20, 0.03, 7).copy()
    rng = np.random.default_rng(1)
    idx = rng.integers(0, n, int(n * dup)); src = rng.integers(0, n, int(n * dup))
    x[idx] = x[src]                                   # exact duplicates: massive ties
    out = {}
    for mode in (0, 1):
        t0 = time.perf_counter(); st, z, stats = fa.linkage(x, mode=mode, ctx=ctx, return_stats=True); dt = time.perf_counter() - t0
        out[mode] = z
        print(n, dup, "mode", mode, "rc", st, round(dt, 3), {k: stats[k] for k in ("rounds", "rescans", "exact_fallback", "windows")})
    h0, h1 = np.sort(out[0][:, 2]), np.sort(out[1][:, 2])
    print("  heights multiset equal:", bool(np.array_equal(h0, h1)), " partitions@0.6 equal:", bool((fa.cut(out[0], n, 0.6) == fa.cut(out[1], n, 0.6)).all()))
