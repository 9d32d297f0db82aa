#!/usr/bin/env python3
"""Throughput of one GPU with several LONG recordings in flight (bench.e2e_many_leg at other shapes).  usage: e2e_many_probe.py R,H [R,H ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context()
for spec in sys.argv[1:]:
    r, h = spec.split(",")
    out = bench.e2e_many_leg(fa, ctx, torch, recordings=int(r), hours_each=float(h), speakers=12)
    print(json.dumps(out))
    ctx.trim()
    torch.cuda.empty_cache()
