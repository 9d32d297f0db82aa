#!/usr/bin/env python3
"""One line per library build: ms per pass of bench.resample_leg's four rate pairs (1 h of audio resident in HBM, HIP events).  The build is chosen by
FLUIDAUDIO_HIP_LIBRARY (scripts/archive/gpu_r4_call25.sh: default policy, nontemporal loads, nontemporal stores, both — builds of a patch that was not kept,
see profiles/r04_resample_nt_ab.txt)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.Context(0)
r = bench.resample_leg(fa, ctx, torch)
print(json.dumps({"lib": os.path.basename(os.environ.get("FLUIDAUDIO_HIP_LIBRARY", "default")),
                  **{k: [round(v["ms_per_pass"], 4), round(v["roofline"]["frac"], 4), v["within_2e-5"]] for k, v in r.items() if isinstance(v, dict)}}), flush=True)
