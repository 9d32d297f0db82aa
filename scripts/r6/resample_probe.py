#!/usr/bin/env python3
"""One rate of the resampler leg of bench.py (one hour of audio -> 16 kHz, three passes) for rocprofv3: FA_PROBE_RATE=<input rate>."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402

PAIRS = {48000: (1, 3), 44100: (160, 441), 22050: (320, 441), 8000: (2, 1), 96000: (1, 6), 88200: (80, 441), 11025: (640, 441), 32000: (1, 2), 37800: (80, 189), 24000: (2, 3)}
rate = int(os.environ["FA_PROBE_RATE"])
up, down = PAIRS[rate]
ctx = fa.default_context()
n = rate * 3600
x = torch.randn(n, device="cuda") * 0.1
n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
y = torch.empty(n_out, device="cuda")
got = C.c_int64()
for _ in range(3):
    ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")
ctx.synchronize()
print("probe done:", rate, up, down, n, n_out)
