"""Resampling kernels: AudioConverter.linearResample bit-exact vs the oracle; polyphase extension vs scipy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ch,frames,rate", [(3, 24000, 48000), (4, 4000, 16000), (5, 800, 8000), (6, 8820, 44100),
                                            (8, 48000, 48000), (3, 1, 48000), (32, 1600, 16000), (3, 40, 4000), (1, 12345, 22050)])
def test_linear_bit_exact(fa, gpu_ctx, oracle_mod, ch, frames, rate):
    x = np.random.default_rng(frames + ch).uniform(-1, 1, (ch, frames)).astype(np.float32)
    ref = oracle_mod.resample_linear(x, rate)
    got = fa.linear_resample(x, rate, ctx=gpu_ctx)
    assert got.size == ref.size
    np.testing.assert_array_equal(got, ref)


def test_linear_reference_known_answers(fa, gpu_ctx):   # AudioConverterTests.swift:571-597, :731-761
    x = np.empty((4, 4000), np.float32)
    x[0], x[1], x[2], x[3] = 0.4, 0.8, -0.4, -0.8
    y = fa.linear_resample(x, 16000, ctx=gpu_ctx)
    assert y.size == 4000 and np.all(np.abs(y) <= 0.001)
    y = fa.linear_resample(np.tile(np.arange(40, dtype=np.float32), (3, 1)), 4000, ctx=gpu_ctx)
    np.testing.assert_array_equal(y[:157], (np.arange(157) * 0.25).astype(np.float32))


@pytest.mark.parametrize("up,down,n", [(1, 3, 48000), (160, 441, 44100), (2, 1, 8000), (1, 1, 1000), (3, 2, 777)])
def test_polyphase_extension_vs_scipy(fa, gpu_ctx, up, down, n):
    from scipy import signal
    rng = np.random.default_rng(n)
    t = np.arange(n) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 300 * t) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    ref = signal.resample_poly(x.astype(np.float64), up, down, window=("kaiser", 5.0))
    got = fa.resample_poly(x, up, down, ctx=gpu_ctx)
    assert got.size == ref.size
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("up,down,n", [(1, 3, 250001), (160, 441, 132300), (2, 1, 40000), (640, 441, 33075), (1, 6, 96000), (1, 1, 5000), (3, 2, 5), (160, 147, 14700),
                                       (1, 2, 100003), (1, 4, 64000), (1, 5, 80007), (1, 6, 300007), (1, 6, 96000), (1, 6, 700), (1, 6, 130), (1, 4, 1000003), (1, 5, 500000), (1, 3, 62), (1, 3, 63), (1, 3, 64), (1, 3, 130), (1, 3, 1000), (1, 2, 45), (1, 5, 200),
                                       # the row-tiled kernel of non-integer ratios (round 4): several tiles, the last one partial, signals too short for a tile
                                       (160, 441, 1000003), (160, 441, 40000), (160, 441, 37000), (320, 441, 300007), (640, 441, 150000), (80, 441, 500000),
                                       (160, 147, 200000), (16, 15, 90000), (8, 7, 50000), (147, 160, 120000), (80, 441, 2000003), (80, 441, 60000), (40, 441, 900000),
                                       # the register-tiled kernel of small interpolation factors (8 / 12 / 24 / 4 / 5.33 / 10.67 kHz -> 16 kHz)
                                       (2, 1, 1000003), (2, 1, 100), (2, 1, 57), (2, 3, 240000), (2, 3, 130), (4, 3, 120001), (4, 1, 40000), (3, 1, 53333), (3, 2, 106667),
                                       (4, 3, 64), (3, 2, 40)])
def test_lds_kernel_equals_simple_kernel(fa, gpu_ctx, switch, up, down, n):
    """The LDS-staged persistent polyphase kernel and the register-tiled decimation kernel (up = 1, down 2 .. 5: interior outputs, the
    edges by the simple kernel) keep the summation order of the one-thread-per-output kernel: identical bits on several rate pairs
    (48k / 44.1k / 8k / 11.025k / 32k / 64k / 80k / 96k -> 16k, 44.1k -> 48k), multi-tile signals, tiny ones, and lengths around the
    point where the first interior group appears."""
    rng = np.random.default_rng(n)
    x = (0.4 * np.sin(2 * np.pi * 440 * np.arange(n) / 16000.0) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    got = fa.resample_poly(x, up, down, ctx=gpu_ctx)
    switch("FA_RESAMPLE_SIMPLE", "1")
    ref = fa.resample_poly(x, up, down, ctx=gpu_ctx)
    switch("FA_RESAMPLE_SIMPLE", None)
    np.testing.assert_array_equal(got, ref)


def test_rows_kernel_actually_serves_the_common_non_integer_pairs(fa, gpu_ctx, switch):
    """44.1 / 22.05 kHz -> 16 kHz go through poly_rows_kernel (not silently through the fallback): with FA_RESAMPLE_NO_ROWS the LDS-staged kernel
    produces the same bits, and on a device-resident hour of audio the row-tiled kernel is the faster of the two by a wide margin."""
    import ctypes as C
    import torch
    for up, down, rate in ((160, 441, 44100), (320, 441, 22050)):
        n = rate * 600
        x = torch.randn(n, device="cuda", dtype=torch.float32) * 0.1
        n_out = fa.lib().fa_resample_poly_frames(n, up, down)
        y = torch.empty(n_out, device="cuda", dtype=torch.float32)
        y2 = torch.empty_like(y)
        got = C.c_int64()
        stream = torch.cuda.ExternalStream(gpu_ctx.stream)

        def run(dst):
            gpu_ctx.check(fa.lib().fa_resample_poly_dev(gpu_ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(dst.data_ptr()), n_out, C.byref(got)), "resample")

        def timed(dst):
            torch.cuda.synchronize()
            run(dst); gpu_ctx.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(3):
                run(dst)
            e1.record(stream)
            gpu_ctx.synchronize()
            return e0.elapsed_time(e1) / 3
        t_rows = timed(y)
        switch("FA_RESAMPLE_NO_ROWS", "1")
        t_lds = timed(y2)
        switch("FA_RESAMPLE_NO_ROWS", None)
        assert torch.equal(y, y2)
        assert t_rows < 0.85 * t_lds, (up, down, t_rows, t_lds)


@pytest.mark.parametrize("tiles", [True, False])
@pytest.mark.parametrize("down,n", [(2, 1000003), (2, 20000), (3, 3000017), (3, 9000), (3, 3300), (4, 640000), (5, 800007), (5, 26000), (6, 3000007), (6, 6200), (6, 700), (12, 4000003), (12, 40000), (3, 3133), (3, 3135), (2, 3113), (2, 3115), (4, 3155), (6, 3193), (12, 3500)])
def test_both_decimation_kernels_equal_simple_kernel(fa, gpu_ctx, switch, tiles, down, n):
    """Round 5: integer decimation through LDS tiles (whole tiles of 256 R outputs; the remainder by the register-tiled kernel and the edges) and, with
    FA_RESAMPLE_NO_DECIM_TILES, by the register-tiled kernel alone: the bits of the one-thread-per-output kernel, from several tiles per workgroup down to
    signals shorter than one tile."""
    if not tiles:
        switch("FA_RESAMPLE_NO_DECIM_TILES", "1")
    rng = np.random.default_rng(n + down)
    x = (0.4 * np.sin(2 * np.pi * 440 * np.arange(n) / 16000.0) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    got = fa.resample_poly(x, 1, down, ctx=gpu_ctx)
    switch("FA_RESAMPLE_SIMPLE", "1")
    ref = fa.resample_poly(x, 1, down, ctx=gpu_ctx)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("form", ["16:8", "32:8", "32:10", "one-tile-per-workgroup"])
@pytest.mark.parametrize("up,down,n", [(160, 441, 1000003), (160, 441, 40000), (160, 441, 15000), (320, 441, 300007), (640, 441, 150000), (80, 189, 200000),
                                       (80, 441, 2000003), (80, 441, 30000)])   # 88.2 kHz: windows of 32 reads — 32-row tiles with ten wavefronts (round 6) / 16-row tiles
def test_every_row_kernel_form_equals_simple_kernel(fa, switch, form, up, down, n):
    """Round 5: the persistent double-buffered row kernels (poly_rows_wide_body: 16-row tiles with two workgroups per CU — the default —, 32-row tiles with one,
    32-row tiles of one phase group with ten wavefronts; FA_RESAMPLE_WIDE picks the form when a context builds its tables) and the one-tile-per-workgroup
    kernel they replaced for 44.1 / 22.05 / 11.025 / 37.8 kHz produce the bits of the one-thread-per-output kernel: several tiles per workgroup, a ragged
    last tile, signals shorter than one tile."""
    if form == "one-tile-per-workgroup":
        switch("FA_RESAMPLE_NO_WIDE", "1")
    else:
        switch("FA_RESAMPLE_WIDE", form)
    rng = np.random.default_rng(n + up)
    x = (0.4 * np.sin(2 * np.pi * 440 * np.arange(n) / 16000.0) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    ctx = fa.Context(0)                     # the tables (and the kernel form) of a pair are fixed when a context first resamples it
    try:
        got = fa.resample_poly(x, up, down, ctx=ctx)
    finally:
        ctx.close()
    switch("FA_RESAMPLE_SIMPLE", "1")
    ref_ctx = fa.Context(0)
    try:
        ref = fa.resample_poly(x, up, down, ctx=ref_ctx)
    finally:
        ref_ctx.close()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("up,down,rate", [(160, 441, 44100), (320, 441, 22050), (80, 441, 88200)])
def test_row_kernels_with_unaligned_device_buffers(fa, gpu_ctx, up, down, rate):
    """fa_resample_poly_dev takes any 4-byte aligned device pointers: an output that is not 16-byte aligned sends the persistent row kernels through their
    piecewise stores, an input that is not through another phase of their unaligned 16-byte global -> LDS requests; both equal the aligned call bit for bit."""
    import ctypes as C
    import torch
    n = rate * 7 + 13
    base = torch.randn(n + 8, device="cuda", dtype=torch.float32) * 0.1
    n_out = fa.lib().fa_resample_poly_frames(n, up, down)
    got = C.c_int64()

    def run(x, y):
        gpu_ctx.check(fa.lib().fa_resample_poly_dev(gpu_ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")
        gpu_ctx.synchronize()
        assert got.value == n_out

    x0 = base[4:4 + n].clone()                               # 16-byte aligned copy of the signal
    y0 = torch.empty(n_out + 8, device="cuda", dtype=torch.float32)
    run(x0, y0[:n_out])
    ref = y0[:n_out].clone()
    for xoff, yoff in ((0, 1), (1, 0), (3, 2), (2, 3)):
        xb = torch.empty(n + 8, device="cuda", dtype=torch.float32)
        xb[xoff:xoff + n] = x0
        yb = torch.full((n_out + 8,), float("nan"), device="cuda", dtype=torch.float32)
        run(xb[xoff:xoff + n], yb[yoff:yoff + n_out])
        assert torch.equal(yb[yoff:yoff + n_out], ref), (xoff, yoff)
        assert torch.isnan(yb[:yoff]).all() and torch.isnan(yb[yoff + n_out:]).all()      # nothing written outside the output


@pytest.mark.parametrize("up,down", [(160, 441), (320, 441)])
def test_rows_kernel_on_non_finite_input(fa, gpu_ctx, switch, up, down):
    """The row-tiled kernel multiplies a register window by a table row whose unused positions hold ZERO taps (the shift of a phase inside its —
    since round 5 shared — window is absorbed by the table).  On finite input that adds +-0 and changes no bit.  An Inf / NaN sample times a zero
    tap is NaN: it reaches every output whose padded window covers it, a few samples more on either side than the true FIR support (where the
    one-output-at-a-time kernel, which skips taps, stays finite).  This is the documented behaviour: outside that neighbourhood the outputs
    equal the simple kernel's bit for bit, inside it they are non-finite in both or only in the row-tiled kernel — never finite-but-different."""
    n = 400000
    rng = np.random.default_rng(up)
    x = (0.1 * rng.standard_normal(n)).astype(np.float32)
    x[n // 2] = np.inf
    x[n // 2 + 50000] = np.nan
    got = fa.resample_poly(x, up, down, ctx=gpu_ctx)
    switch("FA_RESAMPLE_SIMPLE", "1")
    ref = fa.resample_poly(x, up, down, ctx=gpu_ctx)
    switch("FA_RESAMPLE_SIMPLE", None)
    bad_ref, bad_got = ~np.isfinite(ref), ~np.isfinite(got)
    assert bad_ref.sum() > 0 and (bad_got | ~bad_ref).all()                 # wherever the simple kernel is non-finite, so is the row-tiled one
    both_ok = ~bad_got
    np.testing.assert_array_equal(got[both_ok], ref[both_ok])             # never finite-but-different
    extra = bad_got & ~bad_ref
    taps, _ = fa.poly_taps(up, down)
    support = (taps.size + down - 1) // down + 2                          # outputs one input sample can reach through the FIR
    for centre in (n // 2, n // 2 + 50000):
        m = centre * up // down
        idx = np.nonzero(extra[max(0, m - 4 * support):m + 4 * support])[0]
        assert idx.size <= 16 * (up // 160 + 1) + 64                      # the padding widens the neighbourhood by a few outputs per phase, not more
    far = np.ones(got.size, bool)
    for centre in (n // 2, n // 2 + 50000):
        m = centre * up // down
        far[max(0, m - 2 * support):m + 2 * support] = False
    assert np.isfinite(got[far]).all()
