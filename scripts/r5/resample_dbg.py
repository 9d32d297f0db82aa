#!/usr/bin/env python3
"""Round 5 experiment: poly_rows_db_kernel with the arithmetic (FA_DBG=1) or the staging (FA_DBG=2) left out: which side bounds the item period."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402
for name, rate, up, down in (("44.1 kHz -> 16 kHz", 44100, 160, 441), ("22.05 kHz -> 16 kHz", 22050, 320, 441)):
    for dbg in ("0", "6", "7"):
        os.environ.pop("FA_RESAMPLE_NO_WIDE", None); os.environ.pop("FA_DBG", None)
        if dbg == "nodb": os.environ["FA_RESAMPLE_NO_WIDE"] = "1"
        else: os.environ["FA_DBG"] = dbg
        ctx = fa.Context(0)
        stream = torch.cuda.ExternalStream(ctx.stream)
        n = rate * 3600
        x = torch.randn(n, device="cuda", dtype=torch.float32) * 0.1
        n_out = fa.lib().fa_resample_poly_frames(n, up, down)
        y = torch.empty(n_out, device="cuda", dtype=torch.float32)
        got = C.c_int64()
        run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")  # noqa: E731
        torch.cuda.synchronize(); run(); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5): run()
        e1.record(stream); ctx.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"pair": name, "dbg": dbg, "ms": ms, "frac": 4.0 * (n + n_out) / 1e9 / (ms * 1e-3) / 8000}), flush=True)
        ctx.close(); del x, y
