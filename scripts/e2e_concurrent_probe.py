#!/usr/bin/env python3
"""Throughput of ONE GPU with K recordings of 8 h in flight: K host threads, each with its own context (stream) and its own resident inputs,
run the headline step (mel of 1 920 chunks + fa_offline_cluster on 43 200 embeddings) concurrently.  usage: e2e_concurrent_probe.py K [steps]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench  # noqa: E402
import fluidaudio_amd as fa  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = e2e_session(8.0, 12)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_8h.json")))
from e2e_inputs import sha256  # noqa: E402
ctxs = [fa.Context(0) for _ in range(K)]
dev = []
for k in range(K):
    emb = torch.from_numpy(np.ascontiguousarray(s["emb"], np.float32)).cuda()
    rho = torch.from_numpy(np.ascontiguousarray(s["rho"], np.float64)).cuda()
    dev.append((emb, rho))
ok = [True] * K
times = [0.0] * K


def work(k, n):
    for _ in range(n):
        res = fa.cluster_embeddings(dev[k][0], dev[k][1], s["chunks"], s["phi"], ctx=ctxs[k])
        if sha256(np.asarray(res.assignments, np.int32)) != gold["assignments_sha256"]:
            ok[k] = False


for k in range(K):
    work(k, 1)          # warm-up: workspaces
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k, steps)) for k in range(K)]
for t in th:
    t.start()
for t in th:
    t.join()
wall = time.perf_counter() - t0
print(json.dumps({"recordings_in_flight": K, "steps_each": steps, "wall_s": wall, "audio_hours_per_s": K * steps * 8.0 / wall, "s_per_recording": wall / steps,
                  "all_equal_reference_digest": all(ok)}))
