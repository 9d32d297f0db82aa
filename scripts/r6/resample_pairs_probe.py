#!/usr/bin/env python3
"""Polyphase resampler on rate pairs OUTSIDE the bench's six: ten minutes of audio per pair -> 16 kHz, HBM fraction (4 (n_in + n_out) bytes per pass) and the
bits against the one-output-per-thread kernel (FA_RESAMPLE_SIMPLE through fa_debug_set_switch)."""
import ctypes as C
import json
import os
import sys
from math import gcd

os.environ.setdefault("FLUIDAUDIO_HIP_DEBUG_HOOKS", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context()
stream = torch.cuda.ExternalStream(ctx.stream)
rates = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "11025,12000,24000,32000,37800,44056,47250,50000,64000,32768,50400,176400,192000,7350,6000".split(","))]
for rate in rates:
    g = gcd(16000, rate)
    up, down = 16000 // g, rate // g
    n = rate * 600
    x = torch.randn(n, device="cuda") * 0.1
    n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
    y = torch.empty(n_out, device="cuda")
    y0 = torch.empty(n_out, device="cuda")
    got = C.c_int64()
    run = lambda out: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(out.data_ptr()), n_out, C.byref(got)), "resample")  # noqa: E731
    try:
        for _ in range(3):
            run(y)
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            run(y)
        e1.record(stream)
        ctx.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fa.lib().fa_debug_set_switch(b"FA_RESAMPLE_SIMPLE", b"1")
        run(y0)
        ctx.synchronize()
        fa.lib().fa_debug_set_switch(b"FA_RESAMPLE_SIMPLE", None)
        same = bool(torch.equal(y, y0))
        print(json.dumps({"rate": rate, "up": up, "down": down, "ms": round(ms, 4), "frac": round(4.0 * (n + n_out) / (ms * 1e-3) / 8e12, 3), "bits_equal_simple_kernel": same}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"rate": rate, "up": up, "down": down, "error": repr(e)[:200]}), flush=True)
    del x, y, y0
