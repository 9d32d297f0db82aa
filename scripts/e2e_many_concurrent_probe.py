#!/usr/bin/env python3
"""16 recordings of 1 h on one GPU: one batched call (fa_offline_cluster_batch over all 16) against G groups in flight (G host threads, each
with its own context, each one batched call over 16 / G recordings).  usage: e2e_many_concurrent_probe.py [G ...]"""
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fluidaudio_amd as fa  # noqa: E402

R, speakers = 16, 8
phi = np.linspace(2.0, 1.0, 128)
recs, truth = [], []
for r in range(R):
    rng = np.random.default_rng(100 + r)
    n_win = 1800
    n = 3 * n_win
    centers = rng.standard_normal((speakers, 256))
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_win)]).reshape(-1)
    emb = (centers[spk] + 0.03 * rng.standard_normal((n, 256))).astype(np.float32)
    rho = (rng.standard_normal((speakers, 128)) * np.sqrt(phi))[spk] + rng.standard_normal((n, 128))
    recs.append((emb, rho, np.repeat(np.arange(n_win), 3)))
    truth.append(spk)
base_ctx = fa.default_context()
st0, out0 = fa.cluster_embeddings_batch(recs, phi, ctx=base_ctx)
for G in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    ctxs = [fa.Context(0) for _ in range(G)]
    groups = [recs[g::G] for g in range(G)]
    res = [None] * G

    def work(g):
        res[g] = fa.cluster_embeddings_batch(groups[g], phi, ctx=ctxs[g])
    for g in range(G):
        work(g)                                   # warm-up
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    same = all(res[g][1][i].assignments == out0[g + G * i].assignments for g in range(G) for i in range(len(groups[g])))
    print(json.dumps({"groups_in_flight": G, "cluster_s": best, "audio_hours_per_s": R * 1.0 / best, "equals_one_batch": bool(same)}))
    for c in ctxs:
        c.close()
