#!/bin/bash
# The host-side geometry of the round-4 resampler kernels (csrc/resample_geom.h) through the CPU emulation of the kernels' indexing, built with
# AddressSanitizer + UndefinedBehaviorSanitizer: 16 rate pairs x 8 signal lengths (0, 1, a few samples, around one tile, many tiles) for the
# row-tiled kernel, 6 pairs x 8 lengths for the register-tiled interpolation kernel.  No GPU.
cd "$(dirname "$0")/.." || exit 1
g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -o /tmp/libgeom_asan.so tests/cpu/resample_geom_emul.cpp || exit 1
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python - <<'PY'
import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
import fluidaudio_amd as fa
L = C.CDLL('/tmp/libgeom_asan.so')
f32p, i64 = np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_int64
L.rows_emulate.argtypes = [f32p, i64, f32p, i64, C.c_int, C.c_int, i64, i64, f32p, C.POINTER(i64), C.POINTER(i64), np.ctypeslib.ndpointer(np.int32, flags="C")]
L.interp_emulate.argtypes = [f32p, i64, f32p, C.c_int, C.c_int, C.c_int, i64, i64, f32p, C.POINTER(i64), C.POINTER(i64)]
rng = np.random.default_rng(0)
runs = 0
for up, down in ((160, 441), (320, 441), (640, 441), (160, 147), (16, 15), (8, 7), (147, 160), (80, 441), (12, 5), (9, 8), (441, 160), (48, 125), (25, 24), (100, 99), (11, 10), (64, 63)):
    taps, pre = fa.poly_taps(up, down)
    for n in (0, 1, 5, 63, 64 * down + 1, 64 * down * 2 + taps.size, 40000, 130001):
        x = rng.standard_normal(max(n, 1)).astype(np.float32)
        n_out = fa.lib().fa_resample_poly_frames(n, up, down)
        y = np.zeros(max(n_out, 1), np.float32); lo, hi = C.c_int64(), C.c_int64(); info = np.zeros(5, np.int32)
        rc = L.rows_emulate(x, n, taps, taps.size, up, down, pre, n_out, y, C.byref(lo), C.byref(hi), info)
        assert rc in (0, -1), (up, down, n, rc)
        runs += 1
for up, down, nt in ((2, 1, 42), (2, 3, 64), (4, 3, 83), (4, 1, 82), (3, 1, 62), (3, 2, 63)):
    taps, pre = fa.poly_taps(up, down)
    for n in (0, 1, 7, 30, 57, 100, 999, 40000):
        x = rng.standard_normal(max(n, 1)).astype(np.float32)
        n_out = fa.lib().fa_resample_poly_frames(n, up, down)
        y = np.zeros(max(n_out, 1), np.float32); lo, hi = C.c_int64(), C.c_int64()
        rc = L.interp_emulate(x, n, taps, nt, up, down, pre, n_out, y, C.byref(lo), C.byref(hi))
        assert rc == 0, (up, down, n, rc)
        runs += 1
print("sanitized geometry runs:", runs, "- no report")
PY
