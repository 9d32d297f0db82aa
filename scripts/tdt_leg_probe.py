#!/usr/bin/env python3
"""bench.py's TDT leg at several batch sizes / element types (the batch of chunks is the kernel's parallel axis)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench  # noqa: E402
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context(0)
torch.cuda.set_device(0)
CASES = ((1024, "float32"), (1024, "float16"), (4096, "float16"), (4096, "float32"), (8192, "float16"))
if len(sys.argv) > 1:   # e.g. 1024:float32,4096:float32
    CASES = tuple((int(c.split(":")[0]), c.split(":")[1]) for c in sys.argv[1].split(","))
for B, dt in CASES:
    try:
        r = bench.tdt_leg(fa, ctx, torch, B=B, dtype=dt)
        print(json.dumps({"B": B, "dtype": dt, "ms_per_pass": r["ms_per_pass"], "frac": r["roofline"]["frac"], "GBps": r["roofline"]["achieved"],
                          "ok_tables": r["ids_equal_table_walk_all_chunks"], "ok_cpu": r["ids_equal_cpu_restatement_all_chunks"],
                          "rows_mean": r["rows_per_chunk_mean"], "rows_longest": r["rows_of_the_longest_chunk"]}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"B": B, "dtype": dt, "error": repr(e)[:300]}), flush=True)
    torch.cuda.empty_cache()
