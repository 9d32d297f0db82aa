#!/usr/bin/env python3
"""One pass over the kernels that are new in round 4, sized like their bench legs, for rocprofv3 (kernel trace / PMC passes): the row-tiled
and register-tiled resamplers, the one-wavefront-per-chunk TDT walk, the uniform-layout batched AHC round (8 recordings of 8 h: a few hundred
rounds are enough for counters — FA_PROBE_ROUNDS limits nothing in the library, so the whole chain runs), the tiled VBx products."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

ctx = fa.default_context()
what = set((os.environ.get("FA_PROBE", "resample,tdt,uni,vbx")).split(","))
if "resample" in what:
    for rate, up, down in ((44100, 160, 441), (22050, 320, 441), (8000, 2, 1), (48000, 1, 3)):
        n = rate * 3600
        x = torch.randn(n, device="cuda") * 0.1
        n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
        y = torch.empty(n_out, device="cuda")
        got = C.c_int64()
        for _ in range(3):
            ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")
        ctx.synchronize()
        del x, y
if "tdt" in what:
    from fluidaudio_amd.tdt import TdtConfig
    B, U, T, V1 = 1024, 64, 188, 1025
    lg = torch.randn((B, U, T, V1 + 5), device="cuda")
    lg[..., V1 - 1] += 4.0
    for _ in range(3):
        fa.tdt_decode_logits(lg, V1, np.full(B, T, np.int32), config=TdtConfig(blank_id=V1 - 1), max_out=U, ctx=ctx)
    del lg
if "uni" in what:
    k = int(os.environ.get("FA_PROBE_K", "8"))
    probs = []
    for i in range(k):
        x = e2e_session(8.0, 12, seed=5 + i)["emb"].astype(np.float64)
        probs.append(x / np.sqrt((x * x).sum(axis=1, keepdims=True)))
    st, zs = fa.linkage_batch(probs, ctx=ctx)
    assert st == [0] * k
    ctx.trim()
if "vbx" in what:
    s = e2e_session(8.0, 12, sigma=0.041)
    r = fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=ctx)
    assert r.centroids.shape[0] == 12
print("probe done:", sorted(what))
