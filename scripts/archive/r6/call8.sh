#!/bin/bash
# round 6, call 8: 88.2 kHz with three pieces read ahead (the ten-wavefront long-window instance has 170 registers): parity + timing, warm
python -m pytest tests/test_gpu_resample.py -q -p no:cacheprovider 2>&1 | tail -n 2
python - <<'PY'
import ctypes as C, os, sys, json
import torch
sys.path.insert(0, os.getcwd())
import fluidaudio_amd as fa
ctx = fa.Context(0)
stream = torch.cuda.ExternalStream(ctx.stream)
for rep in range(2):
  for rate, up, down in ((44100, 160, 441), (88200, 80, 441)):
    n = rate * 3600
    x = torch.randn(n, device="cuda") * 0.1
    n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
    y = torch.empty(n_out, device="cuda")
    got = C.c_int64()
    run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "r")
    for _ in range(5): run()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10): run()
    e1.record(stream); ctx.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps({"rate": rate, "ms": ms, "frac": 4.0 * (n + n_out) / (ms * 1e-3) / 8e12}))
    del x, y
PY
