"""TDT navigation helpers: the reference's known answers (Tests/FluidAudioTests/ASR/Parakeet/SlidingWindow/TDT/Decoder/
TdtRefactoredComponentsTests.swift:12-195, TdtDecoderChunkTests.swift:140-165) on the oracle AND on the library's host
helpers (no GPU needed), plus hand-traced walks of the control loop (TdtDecoderV3.swift:230-571)."""
import numpy as np
import pytest

B = 8192


def test_initial_time_indices(oracle_mod, fa):
    for fn in (oracle_mod.tdt_initial_time_index, fa.TdtFrameNavigation.calculate_initial_time_indices):
        assert fn(None, 0) == 0 and fn(None, 5) == 5 and fn(10, -5) == 5 and fn(0, 0) == 25 and fn(-10, 5) == 0


def test_navigation_state_and_time_jump(fa):
    nav = fa.TdtFrameNavigation
    assert nav.initialize_navigation_state(10, 100, 80) == (80, 10, 79, True)
    e, s, l, a = nav.initialize_navigation_state(100, 80, 80)
    assert s == 79 and a is False
    assert nav.calculate_final_time_jump(100, 80, True) is None
    assert nav.calculate_final_time_jump(100, 80, False) == 20
    assert nav.calculate_final_time_jump(50, 80, False) == -30
    assert nav.calculate_final_time_jump(143, 140, False) == 3 and nav.calculate_final_time_jump(140, 140, False) == 0


def test_duration_mapping_and_clamp(oracle_mod, fa):
    dm = fa.TdtDurationMapping
    assert [dm.map_duration_bin(i, [1, 2, 3, 4, 5]) for i in range(5)] == [1, 2, 3, 4, 5]
    assert dm.map_duration_bin(3, [1, 1, 2, 3, 5, 8]) == 3 and dm.map_duration_bin(5, [1, 1, 2, 3, 5, 8]) == 8
    for bad in (5, -1):
        with pytest.raises(ValueError, match="Duration bin index out of range"):
            dm.map_duration_bin(bad, [1, 2, 3, 4, 5])
    for fn in (oracle_mod.tdt_clamp_probability, dm.clamp_probability):
        assert fn(0.5) == 0.5 and fn(0.0) == 0.0 and fn(1.0) == 1.0 and fn(-0.5) == 0.0 and fn(-100.0) == 0.0
        assert fn(1.5) == 1.0 and fn(100.0) == 1.0 and fn(float("nan")) == 0.0 and fn(float("inf")) == 0.0 and fn(float("-inf")) == 0.0


def _tables(U, T, fill_tok=B, fill_bin=1):
    return np.full((U, T), fill_tok, np.int32), np.full((U, T), fill_bin, np.int32), np.full((U, T), 0.5, np.float32)


def test_control_loop_hand_traced(oracle_mod):
    # all blank, duration 1: walks to the end, nothing emitted, final time == Teff
    tok, bn, pr = _tables(4, 10)
    r = oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10)
    assert r["count"] == 0 and r["final_time"] == 10 and r["final_u"] == 0 and r["status"] == 0
    # blank with duration bin 0 is forced to advance by 1 (:327-329)
    tok, bn, pr = _tables(4, 10, fill_bin=0)
    assert oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10)["final_time"] == 10
    # token 7 at (u=0, t=2) with duration 2, then blanks: emitted at frame 2 (+ global offset), decoder steps once
    tok, bn, pr = _tables(4, 10)
    tok[0, 2], bn[0, 2], pr[0, 2] = 7, 2, 0.9
    r = oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10, global_offset=162)
    assert r["tokens"].tolist() == [7] and r["timestamps"].tolist() == [164] and r["durations"].tolist() == [2]
    assert abs(float(r["confidences"][0]) - 0.9) < 1e-7 and r["final_u"] == 1 and r["final_time"] == 10
    # warm-up suppression: emitTokensAfterGlobalFrame hides the token but the decoder still steps (:414-431)
    r = oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10, global_offset=162, emit_after=165)
    assert r["count"] == 0 and r["final_u"] == 1
    # duration-0 tokens at one frame: the second emission at the same frame is pushed forward by 1 (:318-323)
    tok, bn, pr = _tables(6, 10)
    tok[0, 0], bn[0, 0] = 5, 0
    tok[1, 0], bn[1, 0] = 6, 0
    r = oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10)
    assert r["tokens"].tolist() == [5, 6] and r["timestamps"].tolist() == [0, 0] and r["durations"].tolist() == [0, 1]
    # a token found exactly when t reaches Teff is not emitted in the main loop (:409), its frame advance is the time jump
    tok, bn, pr = _tables(4, 10)
    tok[0, 9], bn[0, 9] = 9, 4
    r = oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10)
    assert r["count"] == 0 and r["final_time"] == 13
    # ... but the last-chunk flush re-queries the boundary frames (:472-571)
    r = oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10, is_last=True)
    assert r["tokens"].tolist()[:1] == [9] and r["timestamps"][0] == 9
    # guards: encoderSequenceLength <= 1 and start beyond the chunk return before touching timeJump
    assert oracle_mod.tdt_greedy(tok, bn, pr, enc_len=1)["final_time"] is None
    assert oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10, t0=10)["final_time"] is None
    # duration bin out of range -> error (mapDurationBin throws)
    bn[0, 0] = 7
    assert oracle_mod.tdt_greedy(tok, bn, pr, enc_len=10)["status"] == 5
