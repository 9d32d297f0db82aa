import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import fluidaudio_amd as fa
ctx = fa.default_context()
rng = np.random.default_rng(0)
for T, S in [(43200, 12), (43200, 200), (43200, 1000), (20000, 3000)]:
    D = 128
    spk = rng.integers(0, S, T)
    phi = np.linspace(2.0, 1.0, D)
    rho = (rng.standard_normal((S, D)) * np.sqrt(phi))[spk] + rng.standard_normal((T, D))
    v = fa.VBxClustering(phi, ctx=ctx)
    v.refine(rho[:2000], spk[:2000] % 7)
    t0 = time.perf_counter(); out = v.refine(rho, spk); dt = time.perf_counter() - t0
    print(T, S, "seconds", round(dt, 4), "iters", len(out.elbos), "assigned", out.assigned_cluster_count, flush=True)
