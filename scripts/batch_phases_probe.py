#!/usr/bin/env python3
"""Where the wall-clock of fa_offline_cluster_batch goes: prepare (uploads, finite-row filter, widening, normalisation) | linkage of all recordings |
finish (cut, VBx, centroids, assignment, downloads), for 16 x 1 h and 8 x 8 h of the bench sessions."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

ctx = fa.default_context()
for count, hours in ((16, 1.0), (8, 8.0)):
    recs = []
    for k in range(count):
        s = e2e_session(hours, 12, seed=5 + k)
        recs.append((s["emb"], s["rho"], s["chunks"]))
    phi = s["phi"]
    fa.cluster_embeddings_batch(recs, phi, ctx=ctx)
    for rep in range(3):
        t0 = time.perf_counter()
        st, res = fa.cluster_embeddings_batch(recs, phi, ctx=ctx)
        wall = time.perf_counter() - t0
        t = [r.timings for r in res]
        a = res[0].info["ahc"]
        print(json.dumps({"count": count, "hours": hours, "wall_ms": 1e3 * wall, "prepare_ms_max": 1e3 * max(x["inputs_s"] for x in t),
                          "linkage_phase_ms_max": 1e3 * max(x["ahc_s"] for x in t), "vbx_ms_max": 1e3 * max(x["vbx_s"] for x in t), "vbx_ms_mean": 1e3 * float(np.mean([x["vbx_s"] for x in t])),
                          "assign_ms_max": 1e3 * max(x["assign_s"] for x in t), "assign_ms_mean": 1e3 * float(np.mean([x["assign_s"] for x in t])),
                          "total_ms_max": 1e3 * max(x["total_s"] for x in t), "ahc_init_ms": a["init_ms"], "ahc_merge_ms": a["merge_ms"], "rounds": a["rounds"],
                          "audio_hours_per_s": count * hours / wall}), flush=True)
    ctx.trim()
