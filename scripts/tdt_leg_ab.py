#!/usr/bin/env python3
"""bench.tdt_leg for the library named by FLUIDAUDIO_HIP_LIBRARY: one line with ms per pass, the roofline fraction and the two id checks."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.Context(0)
r = bench.tdt_leg(fa, ctx, torch)
print(json.dumps({"lib": os.path.basename(os.environ.get("FLUIDAUDIO_HIP_LIBRARY", "default")), "ms_per_pass": r["ms_per_pass"], "frac": r["roofline"]["frac"],
                  "rows_read": r["rows_read"], "ids_equal_table_walk": r["ids_equal_table_walk_all_chunks"], "ids_equal_cpu": r["ids_equal_cpu_restatement_all_chunks"]}), flush=True)
