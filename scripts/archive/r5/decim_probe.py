#!/usr/bin/env python3
"""Round 5: integer decimation (32 / 48 / 64 / 80 / 96 / 192 kHz -> 16 kHz; 192 kHz has no register-tiled instance: its second line is the LDS-staged kernel) through LDS tiles (poly_decim_tile_kernel) against the register-tiled kernel
(FA_RESAMPLE_NO_DECIM_TILES=1), one hour of device-resident audio per pair; bits against the one-thread-per-output kernel on the first 10 s."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402
out = []
for rate, down in ((32000, 2), (48000, 3), (64000, 4), (80000, 5), (96000, 6), (192000, 12)):
    for form in ("tiles", "registers"):
        os.environ.pop("FA_RESAMPLE_NO_DECIM_TILES", None)
        if form == "registers":
            os.environ["FA_RESAMPLE_NO_DECIM_TILES"] = "1"
        ctx = fa.Context(0)
        stream = torch.cuda.ExternalStream(ctx.stream)
        n = rate * 3600
        x = torch.randn(n, device="cuda", dtype=torch.float32) * 0.1
        n_out = fa.lib().fa_resample_poly_frames(n, 1, down)
        y = torch.empty(n_out, device="cuda", dtype=torch.float32)
        got = C.c_int64()
        run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, 1, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")  # noqa: E731
        torch.cuda.synchronize(); run(); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5): run()
        e1.record(stream); ctx.synchronize()
        ms = e0.elapsed_time(e1) / 5
        n10 = rate * 10
        mine = fa.resample_poly(x[:n10].cpu().numpy(), 1, down, ctx=ctx)
        os.environ["FA_RESAMPLE_SIMPLE"] = "1"
        ref = fa.resample_poly(x[:n10].cpu().numpy(), 1, down, ctx=ctx)
        os.environ.pop("FA_RESAMPLE_SIMPLE")
        rec = {"pair": f"{rate} -> 16000", "form": form, "ms_per_audio_hour": ms, "frac_of_8TBps": 4.0 * (n + n_out) / 1e9 / (ms * 1e-3) / 8000, "bits_equal_simple_kernel_first_10s": bool((ref == mine).all())}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        ctx.close(); del x, y
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5", "decim_probe.json"), "w"), indent=1)
