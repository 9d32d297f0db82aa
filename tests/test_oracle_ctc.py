"""Pins the argmax/CTC oracle with the reference's known-answer tests
(Tests/FluidAudioTests/ASR/LogitsArgmaxTests.swift:9-84,
 Tests/FluidAudioTests/ASR/Parakeet/SlidingWindow/CTC/CtcDecoderTests.swift:64-141)."""
import numpy as np


def test_float32_contiguous_tie_first_index(oracle_mod):
    v = np.array([[0.1, 0.9, -0.3, 0.2, 0.0], [-2.0, -1.0, -0.5, -3.0, -4.0], [7.0, 7.0, 8.0, 8.0, 1.0]], np.float32)
    assert oracle_mod.argmax_rows(v).tolist() == [1, 2, 2]


def test_float16_conversion(oracle_mod):
    v = np.array([[0.25, -0.5, 3.0, 1.5], [-1.0, -0.25, -0.75, -0.125]], np.float16)
    assert oracle_mod.argmax_rows(v).tolist() == [2, 3]


def test_padded_stride_ignores_padding(oracle_mod):
    rows = np.array([[0.5, 0.1, 0.2], [-1.0, -0.2, -0.6], [2.0, 9.0, 3.0]], np.float32)
    st = np.full((3, 4), 1e9, np.float32)
    st[:, :3] = rows
    assert oracle_mod.argmax_rows(st, vocab=3).tolist() == [0, 1, 1]


def test_frame_prefix_and_all_negative(oracle_mod):
    x = np.stack([np.arange(4, dtype=np.float32), -np.arange(4, dtype=np.float32)], 1)
    assert oracle_mod.argmax_rows(x, frames=2).size == 2
    assert oracle_mod.argmax_rows(np.array([[-5.0, -2.0, -9.0]], np.float32)).tolist() == [1]


def test_nan_never_wins_and_degenerate_rows(oracle_mod):
    nan, inf = np.nan, np.inf
    x = np.array([[nan, 1.0, 0.0], [nan, -inf, -inf], [-inf, nan, -inf], [nan, nan, nan], [1.0, nan, 2.0]], np.float32)
    assert oracle_mod.argmax_rows(x).tolist() == [1, 0, 0, 0, 2]


def test_greedy_collapse_cases(oracle_mod):
    L = -100.0
    simple = np.array([[0, L, L], [L, L, 0], [L, 0, L]], np.float32)
    assert oracle_mod.ctc_greedy(simple, 2).tolist() == [0, 1]                      # "hello world"
    rep = np.array([[0, L, L], [0, L, L], [L, 0, L]], np.float32)
    assert oracle_mod.ctc_greedy(rep, 2).tolist() == [0, 1]                         # collapses repeats
    blank_sep = np.array([[0, L], [L, 0], [0, L]], np.float32)
    assert oracle_mod.ctc_greedy(blank_sep, 1).tolist() == [0, 0]                   # blank allows repeats
    allblank = np.array([[L, 0], [L, 0], [L, 0]], np.float32)
    assert oracle_mod.ctc_greedy(allblank, 1).size == 0
    assert oracle_mod.ctc_collapse([], 1).size == 0
    # default blank 1024 with V = 1025
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 1025)).astype(np.float32)
    x[::2, 1024] = 10.0
    ids = oracle_mod.argmax_rows(x)
    exp = [int(i) for t, i in enumerate(ids) if i != 1024 and (t == 0 or ids[t - 1] != i)]
    assert oracle_mod.ctc_greedy(x, 1024).tolist() == exp


def _overload_cases():
    """Frames on which the two ctcGreedyDecode overloads are DEFINED to differ (CtcDecoder.swift:21-31 vs :55-64) and the ones around them."""
    nan, inf = np.nan, np.inf
    return {
        "nan_in_column_0": [[nan, 1.0, 0.0], [0.0, 2.0, 1.0], [nan, 5.0, 9.0], [3.0, nan, 1.0]],
        "all_nan": [[nan, nan, nan], [0.0, 1.0, 0.0], [nan, nan, nan]],
        "neg_inf_seed": [[-inf, -inf, -inf], [-inf, nan, -inf], [-inf, -inf, 1.0]],
        "ragged": [[0.0, 1.0], [0.0, 0.0, 0.0, 7.0], [5.0], [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 8.0], [9.0, 1.0]],
        "empty_between_repeats": [[0.0, 4.0, 0.0], [], [0.0, 4.0, 0.0], [], [], [0.0, 0.0, 4.0], [0.0, 4.0, 0.0]],
        "all_empty": [[], [], []],
        "nan_later_only": [[1.0, nan, 2.0], [0.5, nan, nan]],
    }


def test_rows_overload_restatements_agree_and_known_answers(oracle_mod):
    """The C restatement of the [[Float]] overload against its literal Python transcription, and hand-derived answers."""
    exp = {
        "nan_in_column_0": [0, 1, 0],          # frames -> 0, 1, 0, 0: the NaN seed is never beaten (:25-27); trailing 0 collapses
        "all_nan": [0, 1, 0],
        "neg_inf_seed": [0, 2],
        "ragged": [1, 3, 0, 9, 0],
        "empty_between_repeats": [1, 2, 1],     # a, [], a -> ONE a: the empty frame is skipped before prev is touched (:23)
        "all_empty": [],
        "nan_later_only": [2, 0],
    }
    for name, frames in _overload_cases().items():
        got = oracle_mod.ctc_greedy_rows(frames, blank_id=99).tolist()
        assert got == oracle_mod.ctc_greedy_rows_py(frames, 99).tolist() == exp[name], name


def test_the_two_overloads_differ_exactly_on_a_nan_seed(oracle_mod):
    """[1,T,V] overload (-inf seed, :55-64): a NaN never wins.  [[Float]] overload (frame[0] seed, :24-25): a NaN in column 0 wins.  On
    rectangular input they agree on every frame whose element 0 is not NaN — checked on random matrices with NaN / inf sprinkled in."""
    x = np.array([[np.nan, 1.0, 0.0], [0.0, 2.0, 1.0]], np.float32)
    assert oracle_mod.argmax_rows(x).tolist() == [1, 1]
    assert oracle_mod.ctc_greedy_rows(x, -1, return_frame_ids=True)[1].tolist() == [0, 1]
    rng = np.random.default_rng(5)
    for _ in range(20):
        T, V = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        m = rng.standard_normal((T, V)).astype(np.float32)
        m[rng.random((T, V)) < 0.1] = np.nan
        m[rng.random((T, V)) < 0.05] = -np.inf
        m[rng.random((T, V)) < 0.02] = np.inf
        a = oracle_mod.argmax_rows(m)
        b = oracle_mod.ctc_greedy_rows(m, -1, return_frame_ids=True)[1]
        seed_nan = np.isnan(m[:, 0])
        assert np.array_equal(a[~seed_nan], b[~seed_nan])
        assert (b[seed_nan] == 0).all()
