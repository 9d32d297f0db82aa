#!/usr/bin/env python3
"""The clustering stage of config 5 (8 h -> 43 200 embeddings) twice through fa_offline_cluster; for rocprofv3 --kernel-trace --stats."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402

hours, speakers = 8.0, 12
rng = np.random.default_rng(5)
n_win = int(hours * 3600 / 2)
n = 3 * n_win
centers = rng.standard_normal((speakers, 256))
centers /= np.linalg.norm(centers, axis=1, keepdims=True)
spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_win)]).reshape(-1)
emb = (centers[spk] + 0.03 * rng.standard_normal((n, 256))).astype(np.float32)
phi = np.linspace(2.0, 1.0, 128)
rho = (rng.standard_normal((speakers, 128)) * np.sqrt(phi))[spk] + rng.standard_normal((n, 128))
chunks = np.repeat(np.arange(n_win), 3)
ctx = fa.default_context()
fa.cluster_embeddings(emb[:3000], rho[:3000], chunks[:3000], phi, ctx=ctx)
for _ in range(4):
    t0 = time.perf_counter()
    res = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=ctx)
    print(round(time.perf_counter() - t0, 4), res.timings, res.info['ahc'])
