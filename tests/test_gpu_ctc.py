"""GPU parity: argmax + CTC greedy collapse (HIP, through the C ABI) vs oracle / golden — bit-exact integer outputs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ctc_golden.npz")


def test_committed_golden(fa, gpu_ctx):
    g = np.load(GOLD)
    ids, fids = fa.ctc_greedy_ids_batch(g["logits"], 64, ctx=gpu_ctx, return_frame_ids=True)
    np.testing.assert_array_equal(fids, g["frame_ids"])
    assert [len(i) for i in ids] == g["lens"].tolist()
    np.testing.assert_array_equal(np.concatenate(ids), g["tokens"])


def test_reference_known_answers(fa, gpu_ctx):
    am = fa.LogitsArgmax.argmax_per_frame
    v = np.array([[[0.1, 0.9, -0.3, 0.2, 0.0], [-2.0, -1.0, -0.5, -3.0, -4.0], [7.0, 7.0, 8.0, 8.0, 1.0]]], np.float32)
    assert am(v, 3, ctx=gpu_ctx) == [1, 2, 2]                                   # LogitsArgmaxTests.swift:12-24
    h = np.array([[[0.25, -0.5, 3.0, 1.5], [-1.0, -0.25, -0.75, -0.125]]], np.float16)
    assert am(h, 2, ctx=gpu_ctx) == [2, 3]                                      # :27-39
    st = np.full((1, 3, 4), 1e9, np.float32)
    st[0, :, :3] = [[0.5, 0.1, 0.2], [-1.0, -0.2, -0.6], [2.0, 9.0, 3.0]]
    assert am(st, 3, vocab=3, ctx=gpu_ctx) == [0, 1, 1]                         # :43-66 padded stride
    x = np.stack([np.arange(4, dtype=np.float32), -np.arange(4, dtype=np.float32)], 1)[None]
    assert len(am(x, 2, ctx=gpu_ctx)) == 2                                      # :69-76 frame prefix
    assert am(np.array([[[-5.0, -2.0, -9.0]]], np.float32), 1, ctx=gpu_ctx) == [1]  # :80-84
    L = -100.0
    vocab2 = {0: "▁hello", 1: "▁world"}
    dec = lambda lp, voc, b: fa.ctc_greedy_decode(lp, voc, b, ctx=gpu_ctx)  # noqa: E731
    assert dec([[0, L, L], [L, L, 0], [L, 0, L]], vocab2, 2) == "hello world"    # CtcDecoderTests.swift:64-75
    assert dec([[0, L, L], [0, L, L], [L, 0, L]], vocab2, 2) == "hello world"    # :77-87
    assert dec([[0, L], [L, 0], [0, L]], {0: "▁hello"}, 1) == "hello hello"      # :89-99
    assert dec([[L, 0], [L, 0], [L, 0]], {0: "▁hello"}, 1) == ""                 # :101-111
    assert dec([], {0: "▁hello"}, 1) == ""                                      # :113-117
    arr = np.full((1, 3, 3), L, np.float32)
    arr[0, 0, 0] = arr[0, 1, 2] = arr[0, 2, 1] = 0.0
    assert dec(arr, vocab2, 2) == "hello world"                                  # :121-141 MLMultiArray overload


def test_nan_inf_rows(fa, gpu_ctx, oracle_mod):
    nan, inf = np.nan, np.inf
    x = np.array([[nan, 1.0, 0.0], [nan, -inf, -inf], [-inf, nan, -inf], [nan, nan, nan], [1.0, nan, 2.0], [inf, inf, 0]], np.float32)
    _, fids = fa.ctc_greedy_ids_batch(x, blank_id=-1, ctx=gpu_ctx, return_frame_ids=True)
    assert fids[0].tolist() == oracle_mod.argmax_rows(x).tolist() == [1, 0, 0, 0, 2, 0]


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_unaligned_rows_head_body_tail(fa, gpu_ctx, oracle_mod, dtype):
    """Rows of 1 025 logits (their alignment rotates with the frame index): the winner in the head, the body and the tail of a row, ties between
    them (first index wins), rows of -inf / NaN only (index 0), NaN beside the maximum."""
    T, V = 64, 1025
    x = np.full((1, T, V), -3.0, dtype)
    for t in range(T):
        kind = t % 8
        if kind == 0: x[0, t, t % 4] = 1.0                               # head (or first body element, depending on the row's alignment)
        elif kind == 1: x[0, t, V - 1 - (t % 3)] = 1.0                    # tail
        elif kind == 2: x[0, t, 500 + t] = 1.0                            # body
        elif kind == 3: x[0, t, [2, 700, V - 1]] = 2.0                    # tie between head, body and tail: the first
        elif kind == 4: x[0, t, :] = -np.inf
        elif kind == 5: x[0, t, :] = np.nan
        elif kind == 6: x[0, t, :] = np.nan; x[0, t, V - 2] = -7.0       # the only number sits in the tail
        else: x[0, t, 3] = np.nan; x[0, t, 4] = 0.5; x[0, t, 5] = np.nan
    _, fids = fa.ctc_greedy_ids_batch(x, blank_id=-1, ctx=gpu_ctx, return_frame_ids=True)
    np.testing.assert_array_equal(fids[0], oracle_mod.argmax_rows(x[0]))
    assert fids[0][3] == 2 and fids[0][4] == 0 and fids[0][5] == 0 and fids[0][6] == V - 2 and fids[0][7] == 4


@pytest.mark.parametrize("T,V,W,dtype", [(1, 1, 1, np.float32), (7, 3, 5, np.float32), (300, 1025, 1025, np.float32),
                                          (257, 1024, 1032, np.float32), (2049, 64, 64, np.float32), (5000, 16, 16, np.float32),
                                          (130, 1024, 1024, np.float16), (99, 37, 40, np.float16),
                                          # rows of any alignment through the head + 16-byte body + tail path (round 4): Parakeet CTC's 1 025 logits per frame
                                          # in both dtypes, rows shorter than a vector, strides that rotate the alignment, a SenseVoice-sized vocabulary
                                          (300, 1025, 1025, np.float16), (64, 1027, 1029, np.float32), (50, 9, 9, np.float16), (33, 2, 3, np.float32),
                                          (21, 25055, 25055, np.float32), (19, 8193, 8198, np.float16), (40, 1030, 1031, np.float16)])
def test_random_shapes_vs_oracle(fa, gpu_ctx, oracle_mod, T, V, W, dtype):
    rng = np.random.default_rng(T * 31 + V)
    B = 3
    x = rng.standard_normal((B, T, W)).astype(dtype)
    blank = V - 1
    x[:, :, blank] += 2
    for t in range(0, T - 1, 7):
        x[:, t] = x[:, t + 1]  # inject repeated frames so the collapse has work to do
    x[0, :, 0] = x[0, :, min(1, V - 1)]  # exact ties -> first index must win
    valid = np.array([T, max(T // 2, 0), 0], np.int32)
    ids, fids = fa.ctc_greedy_ids_batch(x, blank, vocab=V, valid_frames=valid, ctx=gpu_ctx, return_frame_ids=True)
    for b in range(B):
        ref_f = oracle_mod.argmax_rows(x[b], vocab=V, frames=int(valid[b]))
        np.testing.assert_array_equal(fids[b, :valid[b]], ref_f)
        np.testing.assert_array_equal(ids[b], oracle_mod.ctc_collapse(ref_f, blank))


def test_full_size_config4_slice_properties(fa, gpu_ctx, oracle_mod):
    """BASELINE config 4 matrices (T=1500, V=1024, blank logit +2, seed 7) on the device-resident path: a 256-matrix slice;
    idempotence-style property: decoding a matrix whose rows were replaced by one-hot(argmax) gives the same ids."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(7)
    B, T, V = 256, 1500, 1024
    x = torch.randn((B, T, V), generator=g, device="cuda", dtype=torch.float32)
    x[:, :, V - 1] += 2.0
    tok = torch.zeros((B, T), dtype=torch.int32, device="cuda")
    lens = torch.zeros(B, dtype=torch.int32, device="cuda")
    fid = torch.zeros((B, T), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fa.ctc_greedy_ids_dev(gpu_ctx, x, V - 1, tok, lens, fid)
    gpu_ctx.synchronize()
    assert torch.equal(fid.long(), x.argmax(dim=2))  # no ties in continuous random data
    for b in (0, 17, 255):
        ref = oracle_mod.ctc_greedy(x[b].cpu().numpy(), V - 1)
        np.testing.assert_array_equal(tok[b, :lens[b]].cpu().numpy(), ref)
    frac_blank = float((fid == V - 1).float().mean())
    assert 0.02 < frac_blank < 0.98
    onehot = torch.zeros_like(x).scatter_(2, fid.long().unsqueeze(-1), 1.0)
    tok2, lens2 = torch.zeros_like(tok), torch.zeros_like(lens)
    fa.ctc_greedy_ids_dev(gpu_ctx, onehot, V - 1, tok2, lens2)
    gpu_ctx.synchronize()
    assert torch.equal(lens, lens2)
    m = torch.arange(T, device="cuda")[None, :] < lens[:, None]
    assert torch.equal(tok[m], tok2[m])


@pytest.mark.parametrize("T,V,dtype,temp,bias", [(50, 1025, "float32", 1.0, 0.0), (37, 1024, "float32", 1.7, 2.5),
                                                  (20, 257, "float16", 0.5, 1.0), (9, 5000, "float32", 1.3, 0.5), (5, 1, "float32", 1.0, 0.0)])
def test_log_softmax_with_temperature_and_blank_bias(fa, gpu_ctx, oracle_mod, T, V, dtype, temp, bias):
    """makeLogProbs / logSoftmax (CtcKeywordSpotter+Inference.swift:350-431); fp32 sums in a different order: 2e-6 abs."""
    import torch
    rng = np.random.default_rng(T * V)
    x = (rng.standard_normal((3, T, V)) * 3).astype(dtype)
    blank = V - 1
    d = fa.ctc_log_probs_dev(gpu_ctx, torch.from_numpy(x).cuda(), temp, bias, blank)
    gpu_ctx.synchronize()
    got = d.cpu().numpy()
    for b in range(3):
        ref = oracle_mod.ctc_log_probs(x[b].astype(np.float32), temp, bias, blank)
        np.testing.assert_allclose(got[b], ref, rtol=0, atol=4e-6 * max(1.0, float(np.abs(ref).max()) / 10))
    p = np.exp(got.astype(np.float64))
    p[:, :, blank] *= np.exp(bias)
    np.testing.assert_allclose(p.sum(-1), 1.0, atol=2e-5)


def _ragged_random(rng, n_frames, max_len, p_empty=0.15, p_nan=0.1):
    frames = []
    for _ in range(n_frames):
        if rng.random() < p_empty:
            frames.append([])
            continue
        n = int(rng.integers(1, max_len + 1))
        f = rng.standard_normal(n).astype(np.float32)
        f[rng.random(n) < p_nan] = np.nan
        if rng.random() < 0.2:
            f[0] = np.nan
        if rng.random() < 0.1:
            f[:] = np.nan
        if rng.random() < 0.1:
            f[rng.integers(0, n)] = np.inf
        frames.append(f)
    return frames


def test_rows_overload_defined_differences(fa, gpu_ctx, oracle_mod):
    """ctcGreedyDecode(logProbs: [[Float]]) (CtcDecoder.swift:15-36): frame[0] seeds the scan (NaN in column 0 -> index 0), frames have
    their own lengths, empty frames do not touch prev — the cases of tests/test_oracle_ctc.py through the C ABI, bit-exact."""
    from test_oracle_ctc import _overload_cases
    for name, frames in _overload_cases().items():
        for blank in (99, 0, 1):
            ids, fids = fa.ctc_greedy_rows([frames], blank, ctx=gpu_ctx, return_frame_ids=True)
            ref, rfids = oracle_mod.ctc_greedy_rows(frames, blank, return_frame_ids=True)
            assert ids[0].tolist() == ref.tolist(), (name, blank)
            assert fids[0].tolist() == rfids.tolist(), (name, blank)
    # the public seam: a list of frames is the [[Float]] overload, an array is the [1, T, V] one — and they differ on a NaN seed
    nan = np.nan
    voc = {0: "a", 1: "b", 2: "c"}
    frames = [[nan, 1.0, 0.0], [0.0, 0.0, 3.0]]
    assert fa.ctc_greedy_decode(frames, voc, 9, ctx=gpu_ctx) == "ac"                       # :24-25 NaN seed wins
    assert fa.ctc_greedy_decode(np.asarray(frames, np.float32), voc, 9, ctx=gpu_ctx) == "bc"   # :55-64 NaN never wins
    assert fa.ctc_greedy_decode([[0.0, 4.0], [], [0.0, 4.0]], voc, 9, ctx=gpu_ctx) == "b"  # :23 empty frame skipped
    assert fa.ctc_greedy_decode([[], []], voc, 9, ctx=gpu_ctx) == ""


def test_rows_overload_random_ragged_batches(fa, gpu_ctx, oracle_mod):
    rng = np.random.default_rng(11)
    # short frames (scalar path), long frames (head + 16-byte body + tail path at every alignment), > one LDS chunk of frames, batches
    for n_frames, max_len, B in ((40, 7, 3), (300, 40, 4), (64, 1100, 2), (5000, 12, 2), (0, 1, 2)):
        utts = [_ragged_random(rng, n_frames, max_len) for _ in range(B)]
        for blank in (0, 3, 10 ** 6):
            ids, fids = fa.ctc_greedy_rows(utts, blank, ctx=gpu_ctx, return_frame_ids=True)
            for u in range(B):
                ref, rfids = oracle_mod.ctc_greedy_rows(utts[u], blank, return_frame_ids=True)
                assert fids[u].tolist() == rfids.tolist()
                assert ids[u].tolist() == ref.tolist()


def test_rows_overload_rectangular_equals_batch_entry_off_the_nan_seed(fa, gpu_ctx):
    """Without NaN in column 0 the two entries are the same function: [T, 1025] logits through both."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((700, 1025)).astype(np.float32)
    x[:, 1024] += 2.0
    a = fa.ctc_greedy_ids_batch(x, 1024, ctx=gpu_ctx)[0]
    b = fa.ctc_greedy_rows([list(x)], 1024, ctx=gpu_ctx)[0]
    assert a.tolist() == b.tolist() and len(a) > 100


def test_rows_overload_argument_contract(fa, gpu_ctx):
    L = fa.lib()
    offs = np.array([0, 3, 2], np.int64)       # decreasing
    vals = np.zeros(4, np.float32)
    tok = np.zeros(2, np.int32)
    n = np.zeros(1, np.int32)
    assert L.fa_ctc_greedy_rows(gpu_ctx.handle, vals.ctypes.data, offs.ctypes.data, 2, None, 1, 0, None, tok.ctypes.data, n.ctypes.data) == 1   # INVALID_ARGUMENT
    assert L.fa_ctc_greedy_rows(gpu_ctx.handle, vals.ctypes.data, offs.ctypes.data, 2, None, 2, 0, None, tok.ctypes.data, n.ctypes.data) == 1   # batch without utt_rows
