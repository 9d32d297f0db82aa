#!/usr/bin/env python3
"""A / B of the rows kernel's LDS budget under the bench leg's own conditions (bench.resample_leg, a fresh context per setting, alternating order)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fluidaudio_amd as fa  # noqa: E402

out = []
for kb in (74, 38, 74, 38, 50, 38):
    os.environ["FA_RESAMPLE_ROWS_LDS_KB"] = str(kb)
    ctx = fa.Context(0)
    r = bench.resample_leg(fa, ctx, torch)
    rec = {"lds_kb": kb, **{k: round(v["ms_per_pass"], 4) for k, v in r.items() if isinstance(v, dict)}}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    ctx.close()
with open(os.path.join(ROOT, "gpurun_out", "summary", "rows_lds_ab.json"), "w") as f:
    json.dump(out, f, indent=1)
