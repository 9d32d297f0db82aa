#!/bin/bash
# round 6, call 6: 88.2 kHz through 32-row tiles with ten wavefronts (one unit each) against the 16-row form: parity, timing, PMC
python -m pytest tests/test_gpu_resample.py -q -p no:cacheprovider 2>&1 | tail -n 3
python - <<'PY'
import ctypes as C, os, sys, time, json
import torch
sys.path.insert(0, os.getcwd())
import fluidaudio_amd as fa
for form in (None, "16:8"):
    fa.lib().fa_debug_set_switch(b"FA_RESAMPLE_WIDE", form.encode() if form else None)
    ctx = fa.Context(0)
    stream = torch.cuda.ExternalStream(ctx.stream)
    for rate, up, down in ((88200, 80, 441), (44100, 160, 441)):
        n = rate * 3600
        x = torch.randn(n, device="cuda") * 0.1
        n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
        y = torch.empty(n_out, device="cuda")
        got = C.c_int64()
        run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "r")
        for _ in range(3): run()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10): run()
        e1.record(stream); ctx.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({"form": form or "default", "rate": rate, "ms": ms, "frac": 4.0 * (n + n_out) / (ms * 1e-3) / 8e12}))
        del x, y
    ctx.close()
PY
RATES="88200" FLUIDAUDIO_HIP_DEBUG_HOOKS=1 bash scripts/r6/resample_pmc.sh 2>&1 | grep -v "rc=0" | cut -c1-1200
