#!/usr/bin/env python3
"""K recordings of 8 h through ONE fa_offline_cluster_batch call: uniform batches side by side (FA_AHC_UNI_GROUPS = 1 .. 4) by K.
Round 5 measured three batches for K = 8 on the linkage alone (profiles/r05_groups_probe.txt); this is the whole call on the round-6 tree."""
import json
import os
import sys
import time

os.environ.setdefault("FLUIDAUDIO_HIP_DEBUG_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa  # noqa: E402
from e2e_inputs import e2e_session  # noqa: E402

ks = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8", "12"])]
groups = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2", "3", "4"])]   # 0: the library's own choice
dev = "--dev" in sys.argv   # inputs resident in HBM (fa_offline_cluster_batch_dev)
ctx = fa.default_context()
sessions = {}
for k in range(max(ks)):
    s = e2e_session(8.0, 12, seed=5 + k)
    sessions[k] = (s["emb"], s["rho"], s["chunks"])
phi = s["phi"]
if dev:
    import torch
    sessions = {k: (torch.from_numpy(e).cuda(), torch.from_numpy(r).cuda(), c) for k, (e, r, c) in sessions.items()}
    torch.cuda.synchronize()
for K in ks:
    recs = [sessions[k] for k in range(K)]
    for g in groups:
        assert fa.lib().fa_debug_set_switch(b"FA_AHC_UNI_GROUPS", str(g).encode() if g else None) == 0
        ctx.trim()
        fa.cluster_embeddings_batch(recs, phi, ctx=ctx)
        walls = []
        for rep in range(3):
            t0 = time.perf_counter()
            st, res = fa.cluster_embeddings_batch(recs, phi, ctx=ctx)
            walls.append(time.perf_counter() - t0)
        a = res[0].info["ahc"]
        t = [r.timings for r in res]
        print(json.dumps({"K": K, "groups": g, "resident": dev, "wall_ms": [round(1e3 * w, 1) for w in walls], "audio_hours_per_s": round(K * 8.0 / min(walls), 1),
                          "prepare_ms_max": round(1e3 * max(x["inputs_s"] for x in t), 1), "linkage_ms_max": round(1e3 * max(x["ahc_s"] for x in t), 1),
                          "ahc_init_ms": round(a["init_ms"], 1), "ahc_merge_ms": round(a["merge_ms"], 1), "rounds": a["rounds"]}), flush=True)
    fa.lib().fa_debug_set_switch(b"FA_AHC_UNI_GROUPS", None)
