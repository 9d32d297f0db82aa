"""CPU check of the host-side geometry of the round-4 resampler kernels (csrc/resample_geom.h): tests/cpu/resample_geom_emul.cpp replays the
indexing of poly_rows_kernel (row-tiled, non-integer ratios) and poly_interp_kernel (register-tiled, small interpolation factors) with the tables
the library builds — every staged / read / written index range-checked, every output of the covered range written once — and the values equal a
plain one-output-at-a-time evaluation bit for bit, which in turn equals scipy.signal.resample_poly at 2e-5.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "cpu", "libresample_geom_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "cpu", "resample_geom_emul.cpp")], check=True)
    L = C.CDLL(so)
    f32p, i64 = np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_int64
    L.poly_simple.argtypes = [f32p, i64, f32p, i64, C.c_int, C.c_int, i64, f32p, i64, i64]
    L.poly_simple.restype = None
    L.rows_emulate.argtypes = [f32p, i64, f32p, i64, C.c_int, C.c_int, i64, i64, f32p, C.POINTER(i64), C.POINTER(i64), np.ctypeslib.ndpointer(np.int32, flags="C"), C.c_int, i64, C.c_int]
    L.interp_emulate.argtypes = [f32p, i64, f32p, C.c_int, C.c_int, C.c_int, i64, i64, f32p, C.POINTER(i64), C.POINTER(i64)]
    L.decim_tiles_emulate.argtypes = [f32p, i64, f32p, C.c_int, i64, f32p, C.POINTER(i64), C.POINTER(i64)]
    return L


def signal_of(n, seed):
    rng = np.random.default_rng(seed)
    return (0.4 * np.sin(2 * np.pi * 440 * np.arange(n) / 16000.0) + 0.1 * rng.standard_normal(n)).astype(np.float32)


@pytest.mark.parametrize("up,down,n", [(160, 441, 132300), (160, 441, 40000), (160, 441, 37000), (160, 441, 28223), (320, 441, 90007), (640, 441, 60000), (160, 147, 70000),
                                       (16, 15, 9000), (8, 7, 3000), (147, 160, 50000), (80, 441, 100000), (12, 5, 4000), (9, 8, 1000), (441, 160, 30000)])
@pytest.mark.parametrize("share_max,budget,rows", [(4, 0, 64), (2, 0, 64), (1, 0, 64), (4, 1 << 30, 32), (4, 1 << 30, 16), (4, 73728, 32)])
def test_rows_kernel_indexing_on_the_library_geometry(fa, emul, up, down, n, share_max, budget, rows):
    """budget 2^30, rows 32 / 16: the geometry of the wide kernels (every phase of a tile of `rows` rows in one item); budget 72 KB, rows 32: their grouped
    form (phase groups whose 32 rows of at most 288 floats fit two buffers twice per CU).  share_max: the most consecutive phases that may read one register window (round 5; 1 = every phase its own window, the round-4 reads)."""
    from scipy import signal
    taps, pre = fa.poly_taps(up, down)
    x = signal_of(n, n)
    n_out = fa.lib().fa_resample_poly_frames(n, up, down)
    y = np.full(n_out, np.nan, np.float32)
    lo, hi = C.c_int64(), C.c_int64()
    info = np.zeros(6, np.int32)
    rc = emul.rows_emulate(x, n, taps, taps.size, up, down, pre, n_out, y, C.byref(lo), C.byref(hi), info, share_max, budget, rows)
    assert rc in (0, -1), (rc, info.tolist())
    if rc == -1:
        assert (taps.size + up - 1) // up + 3 > 128 or up < 8, "only pairs whose phase does not fit a table row may be refused"
        return
    ref = np.zeros(n_out, np.float32)
    emul.poly_simple(x, n, taps, taps.size, up, down, pre, ref, 0, n_out)
    np.testing.assert_allclose(ref, signal.resample_poly(x.astype(np.float64), up, down, window=("kaiser", 5.0)), rtol=0, atol=2e-5)
    if hi.value > lo.value:
        assert lo.value % 4 == 0 and info[2] % 4 == 0 and (info[3] // 4) % 2 == 1 and info[3] * 256 <= 150 * 1024
        assert info[5] in (1, 2, 4) and info[5] <= share_max
        if (up, down) == (160, 441):
            assert (info[5], info[0]) == (min(share_max, 2), 16)             # 44.1 kHz: two phases share the 16 reads one phase needed
        if (up, down) == (320, 441) and share_max == 4:
            assert info[5] == 4 and info[0] <= 10                            # 22.05 kHz: four phases share <= 10 reads (8 each before)
        np.testing.assert_array_equal(y[lo.value:hi.value], ref[lo.value:hi.value])
        assert np.isnan(y[:lo.value]).all() and np.isnan(y[hi.value:]).all()      # the edges belong to poly_kernel
        assert hi.value - lo.value > 0.5 * n_out or n < 64 * down * 3              # long signals are mostly served by the tiles
    else:
        assert info[4] == 0


@pytest.mark.parametrize("up,down,nt,n", [(2, 1, 42, 8000), (2, 1, 42, 57), (2, 1, 42, 30), (2, 3, 64, 24000), (2, 3, 64, 130), (4, 3, 83, 12001), (4, 1, 82, 4000), (3, 1, 62, 5333),
                                          (3, 2, 63, 10667), (4, 3, 83, 64), (3, 2, 63, 40)])
def test_interp_kernel_indexing_on_the_library_geometry(fa, emul, up, down, nt, n):
    taps, pre = fa.poly_taps(up, down)
    assert taps.size == nt                                                       # the instantiated tap counts of resample.hip
    x = signal_of(n, n + 1)
    n_out = fa.lib().fa_resample_poly_frames(n, up, down)
    y = np.full(n_out, np.nan, np.float32)
    lo, hi = C.c_int64(), C.c_int64()
    rc = emul.interp_emulate(x, n, taps, nt, up, down, pre, n_out, y, C.byref(lo), C.byref(hi))
    assert rc == 0, rc
    ref = np.zeros(n_out, np.float32)
    emul.poly_simple(x, n, taps, taps.size, up, down, pre, ref, 0, n_out)
    np.testing.assert_array_equal(y[lo.value:hi.value], ref[lo.value:hi.value])
    assert np.isnan(y[:lo.value]).all() and np.isnan(y[hi.value:]).all()
    if n >= 1000:
        assert hi.value - lo.value > 0.9 * n_out


@pytest.mark.parametrize("down,n", [(2, 40000), (2, 12000), (3, 100003), (3, 9000), (3, 3300), (4, 64000), (5, 80007), (5, 25000), (6, 300007), (6, 6200), (6, 700), (12, 400003), (12, 3500),
                                    # a last tile whose last input sits 1 .. 3 samples before the end of the signal (n = (TO t + 20) down + {1, 2, 3})
                                    (3, 3133), (3, 3134), (3, 3135), (2, 3113), (2, 3114), (2, 3115), (3, 6205), (4, 3155), (5, 5221), (6, 3193), (12, 3314)])
def test_decim_tile_kernel_indexing(fa, emul, down, n):
    """Round 5: integer decimation through LDS tiles (poly_decim_tile_kernel): the tiles the host launches, every staged piece inside the signal or clamped and
    never read where clamped, every window inside the buffer, every output of the tiles written once; values = the one-output-at-a-time evaluation bit for bit."""
    taps, pre = fa.poly_taps(1, down)
    assert taps.size == 21 * down + 1 and pre == 11
    x = signal_of(n, n + down)
    n_out = fa.lib().fa_resample_poly_frames(n, 1, down)
    y = np.full(n_out, np.nan, np.float32)
    lo, hi = C.c_int64(), C.c_int64()
    rc = emul.decim_tiles_emulate(x, n, taps, down, n_out, y, C.byref(lo), C.byref(hi))
    assert rc == 0, rc
    ref = np.zeros(n_out, np.float32)
    emul.poly_simple(x, n, taps, taps.size, 1, down, pre, ref, 0, n_out)
    assert (hi.value - lo.value) % 256 == 0
    if n // down > 3000 and down != 12 and n > 10000:
        assert hi.value > lo.value                                          # at least one tile
    np.testing.assert_array_equal(y[lo.value:hi.value], ref[lo.value:hi.value])
    assert np.isnan(y[:lo.value]).all() and np.isnan(y[hi.value:]).all()   # the rest belongs to the register-tiled kernel and the edges
