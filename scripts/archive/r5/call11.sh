#!/bin/bash
# round 5, call 11: row-tiled resampler with a register cap (three workgroups per CU) x LDS budget
cd "$GRAFT_REPO_ROOT" || exit 1
cd fluidaudio_amd/csrc && mkdir -p variants
for w in 6 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "-DFA_ROWS_ATTR=__attribute__((amdgpu_waves_per_eu($w,$w)))" -c resample.hip -o variants/resample_w$w.o 2>&1 | tail -2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_rows_w$w.so ctx.o pool.o mel.o ctc.o beam.o tdt.o ahc.o vbx.o post.o kmeans.o variants/resample_w$w.o formats.o offline.o
done
cd "$GRAFT_REPO_ROOT"
cat > /tmp/rs_ab.py <<'PY'
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import fluidaudio_amd as fa
for name, rate, up, down in (("44.1", 44100, 160, 441), ("22.05", 22050, 320, 441)):
    ctx = fa.Context(0)
    stream = torch.cuda.ExternalStream(ctx.stream)
    n = rate * 3600
    x = torch.randn(n, device="cuda") * 0.1
    n_out = fa.lib().fa_resample_poly_frames(n, up, down)
    y = torch.empty(n_out, device="cuda")
    got = C.c_int64()
    run = lambda: ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "rs")
    run(); ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(5): run()
    e1.record(stream); ctx.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"lib": os.path.basename(os.environ.get("FLUIDAUDIO_HIP_LIBRARY", "default")), "lds_kb": os.environ.get("FA_RESAMPLE_ROWS_LDS_KB", "auto"), "rate": name, "ms": ms,
                      "frac": 4.0 * (n + n_out) / 1e9 / (ms * 1e-3) / 8000.0, "checksum": float(y[::1000].double().sum())}), flush=True)
    ctx.close()
PY
for lib in default variants/lib_rows_w5.so variants/lib_rows_w6.so; do
  for kb in auto 50 38; do
    if [ "$lib" = default ]; then unset FLUIDAUDIO_HIP_LIBRARY; else export FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/fluidaudio_amd/csrc/$lib; fi
    if [ "$kb" = auto ]; then unset FA_RESAMPLE_ROWS_LDS_KB; else export FA_RESAMPLE_ROWS_LDS_KB=$kb; fi
    python /tmp/rs_ab.py 2>&1 | grep -v amdgpu.ids
  done
done
