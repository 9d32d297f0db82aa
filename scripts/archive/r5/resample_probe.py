#!/usr/bin/env python3
"""Round 5: the row-tiled polyphase kernel with shared register windows + DPP-broadcast taps, per FA_RESAMPLE_ROWS_SHARE (1 = one window per phase,
the reads of round 4), on one hour of device-resident audio; checks bits against the one-thread-per-output kernel on the first 10 s."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402

out = []
for name, rate, up, down in (("44.1 kHz -> 16 kHz", 44100, 160, 441), ("22.05 kHz -> 16 kHz", 22050, 320, 441), ("11.025 kHz -> 16 kHz", 11025, 640, 441), ("48 kHz -> 16 kHz", 48000, 1, 3),
                             ("8 kHz -> 16 kHz", 8000, 2, 1), ("96 kHz -> 16 kHz", 96000, 1, 6), ("88.2 kHz -> 16 kHz", 88200, 80, 441)):
    for share in ("4", "2", "1"):
        if share != "4" and up < 8:
            continue
        os.environ["FA_RESAMPLE_ROWS_SHARE"] = share
        ctx = fa.Context(0)                                   # the tables of a pair are built once per context: a fresh one per setting
        stream = torch.cuda.ExternalStream(ctx.stream)
        n = rate * 3600
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
        # bits against the simple kernel on the first 10 s
        n10 = rate * 10
        os.environ["FA_RESAMPLE_SIMPLE"] = "1"
        c2 = fa.Context(0)
        ref = fa.resample_poly(x[:n10].cpu().numpy(), up, down, ctx=c2)
        os.environ.pop("FA_RESAMPLE_SIMPLE")
        mine = fa.resample_poly(x[:n10].cpu().numpy(), up, down, ctx=ctx)
        gb = 4.0 * (n + n_out) / 1e9
        rec = {"pair": name, "share_max": share, "ms_per_audio_hour": ms, "frac_of_8TBps": gb / (ms * 1e-3) / 8000.0, "bits_equal_simple_kernel_first_10s": bool((ref == mine).all())}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        c2.close(); ctx.close()
        del x, y
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5", "resample_probe.json"), "w"), indent=1)
