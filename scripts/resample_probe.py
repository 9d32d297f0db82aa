#!/usr/bin/env python3
"""Polyphase resampler on device-resident signals: HBM roofline (algorithmic bytes = 4 (n_in + n_out)) for common rate pairs."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context()
stream = torch.cuda.ExternalStream(ctx.stream)
out = []
for name, rate, up, down in (("48 kHz -> 16 kHz", 48000, 1, 3), ("44.1 kHz -> 16 kHz", 44100, 160, 441), ("8 kHz -> 16 kHz", 8000, 2, 1), ("22.05 kHz -> 16 kHz", 22050, 320, 441)):
    for simple in (False, True):
        if simple:
            os.environ["FA_RESAMPLE_SIMPLE"] = "1"
        else:
            os.environ.pop("FA_RESAMPLE_SIMPLE", None)
        n = rate * 3600            # one hour of audio
        x = torch.randn(n, device="cuda", dtype=torch.float32) * 0.1
        n_out = fa.lib().fa_resample_poly_frames(n, up, down)
        y = torch.empty(n_out, device="cuda", dtype=torch.float32)
        got = C.c_int64()
        run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")  # noqa: E731
        torch.cuda.synchronize()
        run(); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            run()
        e1.record(stream)
        ctx.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gb = 4.0 * (n + n_out) / 1e9
        out.append({"pair": name, "kernel": "one thread per output (global memory)" if simple else "LDS-staged persistent", "ms_per_audio_hour": ms, "audio_hours_per_s": 1e3 / ms,
                    "GBps": gb / (ms * 1e-3), "frac_of_8TBps": gb / (ms * 1e-3) / 8000.0})
        del x, y
print(json.dumps({"resample_poly": out, "note": "includes the host-side tap design + 35 KB tap upload of every call (~0.1 ms)"}))
