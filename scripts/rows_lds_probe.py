#!/usr/bin/env python3
"""poly_rows_kernel against the LDS budget of a phase group (FA_RESAMPLE_ROWS_LDS_KB: 74 = two workgroups per CU, 50 = three, 38 = four): one hour of
44.1 / 22.05 kHz audio resident in HBM, one fresh context per setting (the tables of a rate pair are built on the first call of a context)."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402

out = []
for kb in (74, 60, 50, 38, 30):
    os.environ["FA_RESAMPLE_ROWS_LDS_KB"] = str(kb)
    ctx = fa.Context(0)
    stream = torch.cuda.ExternalStream(ctx.stream)
    for rate, up, down in ((44100, 160, 441), (22050, 320, 441)):
        n = rate * 3600
        x = torch.randn(n, device="cuda") * 0.1
        n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
        y = torch.empty(n_out, device="cuda")
        got = C.c_int64()
        run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")  # noqa: E731
        torch.cuda.synchronize()
        for _ in range(3):
            run()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            run()
        e1.record(stream)
        ctx.synchronize()
        ms = e0.elapsed_time(e1) / 10
        rec = {"lds_kb": kb, "rate": rate, "ms_per_pass": ms, "frac_of_8TBps": 4.0 * (n + n_out) / (ms * 1e-3) / 8e12, "checksum": float(y[::997].double().sum())}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del x, y
    ctx.close()
with open(os.path.join(ROOT, "gpurun_out", "summary", "rows_lds_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
