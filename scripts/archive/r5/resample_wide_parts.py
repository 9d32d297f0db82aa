#!/usr/bin/env python3
"""Round 5: the wide row kernels (FA_RESAMPLE_WIDE = rows:wavefronts) in PARTS (FA_RESAMPLE_WIDE_PART: 1 staging only, 2 no staging, 3 arithmetic + stores
without the period wait / barrier, 4 arithmetic alone; 0 = the kernel) next to poly_rows_kernel ("nodb"), one hour of audio per pair."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402
out = []
for name, rate, up, down in (("44.1 kHz -> 16 kHz", 44100, 160, 441), ("22.05 kHz -> 16 kHz", 22050, 320, 441), ("11.025 kHz -> 16 kHz", 11025, 640, 441)):
    for dbg in ("16:8:0", "16:8:1", "16:8:2", "16:8:3", "16:8:4", "32:8:0", "32:8:1", "32:8:4", "32:10:0", "32:10:1", "32:10:4", "nodb"):
        os.environ.pop("FA_RESAMPLE_NO_WIDE", None); os.environ.pop("FA_RESAMPLE_WIDE_PART", None)
        if dbg == "nodb": os.environ["FA_RESAMPLE_NO_WIDE"] = "1"
        else: os.environ["FA_RESAMPLE_WIDE"], os.environ["FA_RESAMPLE_WIDE_PART"] = dbg.rsplit(":", 1)
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
        rec = {"pair": name, "rows:wavefronts:part": dbg, "ms_per_audio_hour": ms, "frac_of_8TBps": 4.0 * (n + n_out) / 1e9 / (ms * 1e-3) / 8000}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        ctx.close(); del x, y
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5", "resample_wide_steps.json"), "w"), indent=1)
