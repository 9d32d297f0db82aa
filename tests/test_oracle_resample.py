"""Known-answer cases of AudioConverter.linearResample, restated from the reference's XCTest file
(Tests/FluidAudioTests/Shared/AudioConverterTests.swift:546-761) against the CPU oracle."""
import numpy as np


def _approx_count(got, expected, tol=0.01):
    assert abs(got - expected) <= max(1, int(expected * tol))


def test_three_channel_48k_to_16k_length(oracle_mod):   # :546-569
    n = 24000
    t = np.arange(n, dtype=np.float32) / np.float32(48000)
    x = np.stack([np.sin(2 * np.pi * f * t) for f in (440, 880, 1320)]).astype(np.float32)
    y = oracle_mod.resample_linear(x, 48000)
    _approx_count(y.size, int(n * 16000 / 48000))


def test_four_channel_mixdown_no_resampling(oracle_mod):   # :571-597
    x = np.empty((4, 4000), np.float32)
    x[0], x[1], x[2], x[3] = 0.4, 0.8, -0.4, -0.8
    y = oracle_mod.resample_linear(x, 16000)
    assert y.size == 4000 and np.all(np.abs(y) <= 0.001)


def test_five_channel_upsampling_preserves_average(oracle_mod):   # :599-626
    x = np.stack([np.full(800, c * 0.2, np.float32) for c in range(5)])
    y = oracle_mod.resample_linear(x, 8000)
    _approx_count(y.size, 1600)
    assert abs(float(y.sum() / np.float32(y.size)) - 0.4) <= 0.01


def test_six_channel_ramp_44k(oracle_mod):   # :628-659
    n = 8820
    ramp = (np.arange(n, dtype=np.float32) / np.float32(n))
    y = oracle_mod.resample_linear(np.tile(ramp, (6, 1)), 44100)
    _approx_count(y.size, int(n * 16000 / 44100))
    assert y[0] < 0.01 and y[-1] > 0.99


def test_edge_cases(oracle_mod):   # :685-729
    assert oracle_mod.resample_linear(np.full((3, 1), 0.5, np.float32), 48000).size == 0   # 1 frame, 3:1 -> 0 samples
    y = oracle_mod.resample_linear(np.ones((32, 1600), np.float32), 16000)
    assert y.size == 1600 and np.all(np.abs(y - 1.0) <= 0.001)


def test_interpolation_accuracy(oracle_mod):   # :731-761
    x = np.tile(np.arange(40, dtype=np.float32), (3, 1))
    y = oracle_mod.resample_linear(x, 4000)
    _approx_count(y.size, 160)
    assert np.all(np.abs(np.diff(y)) < 0.5)
    # 4x upsampling of a ramp: exact quarter steps until the last input sample, then the sample itself (:431-432)
    np.testing.assert_array_equal(y[:157], (np.arange(157) * 0.25).astype(np.float32))
    assert y[156] == 39.0 and np.all(y[157:] == 39.0)


def test_poly_taps_follow_scipy_specification(fa):
    """The polyphase extension's FIR = scipy.signal.resample_poly's (host computation of the library, no GPU)."""
    from scipy import signal
    for up, down in ((1, 3), (160, 441), (2, 1), (3, 2)):
        taps, pre = fa.poly_taps(up, down)
        mx = max(up, down)
        half = 10 * mx
        h = signal.firwin(2 * half + 1, 1.0 / mx, window=("kaiser", 5.0)) * up
        pre_pad = down - half % down
        assert taps.size == h.size + pre_pad and pre == (half + pre_pad) // down
        assert np.all(taps[:pre_pad] == 0)
        np.testing.assert_allclose(taps[pre_pad:], h.astype(np.float32), rtol=0, atol=2e-7 * up)


def test_output_lengths_need_no_gpu(fa, oracle_mod):
    """The frame-count entries are host arithmetic: linear = the restatement's count (AudioConverter.swift:396-400: Int(Double(frames) *
    ratio) with ratio = target / source as Double), polyphase = ceil(n * up / down) as scipy.signal.resample_poly defines it."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(0)
    for _ in range(300):
        frames = int(rng.integers(0, 200000))
        sr = float(rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000, 96000, 12345.6]))
        tr = float(rng.choice([16000, 8000, 24000, 44100]))
        assert fa.lib().fa_resample_linear_frames(frames, sr, tr) == oracle_mod.lib().fa_oracle_resample_linear_frames(frames, sr, tr), (frames, sr, tr)
    for n in (0, 1, 2, 7, 160, 161, 44100, 48000, 100003):
        for up, down in ((1, 3), (1, 2), (160, 441), (2, 3), (3, 1), (1, 1), (1, 5), (320, 441)):
            want = len(resample_poly(np.zeros(n), up, down)) if n else 0
            assert fa.lib().fa_resample_poly_frames(n, up, down) == want, (n, up, down)
