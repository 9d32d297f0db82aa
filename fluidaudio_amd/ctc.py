"""Host-side mirror of ``LogitsArgmax.argmaxPerFrame`` and the greedy half of ``ctcGreedyDecode``
(reference: Sources/FluidAudio/ASR/Shared/LogitsArgmax.swift:16-55,
Sources/FluidAudio/ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:15-70,292-297) over the HIP C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

SENTENCEPIECE_WORD_BOUNDARY = "▁"  # ASRConstants.sentencePieceWordBoundary


def _as_matrix_batch(logits) -> np.ndarray:
    x = np.asarray(logits)
    if x.dtype not in (np.float32, np.float16):
        x = x.astype(np.float32)
    if x.ndim == 2:
        x = x[None]
    assert x.ndim == 3, "logits must be [T,V] or [B,T,V]"
    return np.ascontiguousarray(x)


def ctc_greedy_ids_batch(logits, blank_id: int, vocab: int | None = None, valid_frames=None, ctx: L.Context | None = None,
                         return_frame_ids: bool = False):
    """[B,T,W] host logits -> list of collapsed id arrays (and optionally the [B,T] per-frame argmax)."""
    ctx = ctx or L.default_context()
    x = _as_matrix_batch(logits)
    B, T, W = x.shape
    V = W if vocab is None else vocab
    dtype = L.DTYPE_F16 if x.dtype == np.float16 else L.DTYPE_F32
    tok = np.zeros((B, max(T, 1)), np.int32)
    lens = np.zeros(B, np.int32)
    fids = np.zeros((B, max(T, 1)), np.int32) if return_frame_ids else None
    vf = None if valid_frames is None else np.ascontiguousarray(valid_frames, np.int32)
    if B > 0:
        ctx.check(L.lib().fa_ctc_greedy_batch(ctx.handle, x.ctypes.data if x.size else None, dtype, B, T, V, W, T * W,
                                              None if vf is None else vf.ctypes.data, blank_id,
                                              None if fids is None else fids.ctypes.data, tok.ctypes.data,
                                              lens.ctypes.data), "fa_ctc_greedy_batch")
    out = [tok[b, :lens[b]].copy() for b in range(B)]
    return (out, fids[:, :T]) if return_frame_ids else out


def ctc_greedy_ids_dev(ctx: L.Context, d_logits, blank_id: int, d_token_ids, d_token_lens, d_frame_ids=None,
                       d_valid_frames=None, vocab: int | None = None, order: bool = True):
    """Device-resident batch: d_logits torch CUDA tensor [B,T,W] (fp32/fp16, contiguous).  Enqueues on ctx.stream, ordered
    against torch's current stream (Context.torch_ordered) unless order=False."""
    import torch
    B, T, W = d_logits.shape
    V = W if vocab is None else vocab
    dtype = L.DTYPE_F16 if d_logits.dtype == torch.float16 else L.DTYPE_F32
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with ctx.torch_ordered(order):
        ctx.check(L.lib().fa_ctc_greedy_batch_dev(ctx.handle, p(d_logits), dtype, B, T, V, W, T * W, p(d_valid_frames),
                                                  blank_id, p(d_frame_ids), p(d_token_ids), p(d_token_lens)),
                  "fa_ctc_greedy_batch_dev")


class LogitsArgmax:
    """enum LogitsArgmax (LogitsArgmax.swift:13)."""

    @staticmethod
    def argmax_per_frame(logits, frames: int, vocab: int | None = None, ctx: L.Context | None = None) -> list[int]:
        """argmaxPerFrame(logits:frames:) (:16-55): logits [1,T,W] (W = row stride >= vocab) or [T,W]."""
        x = np.asarray(logits)
        if x.ndim == 3:
            x = x[0]
        frames = min(frames, x.shape[0])
        if frames <= 0:
            return []
        _, fids = ctc_greedy_ids_batch(x[:frames], blank_id=-1, vocab=vocab, ctx=ctx, return_frame_ids=True)
        return np.asarray(fids[0]).tolist()


def decode_ctc_token_ids(ids, vocabulary: dict[int, str]) -> str:
    """decodeCtcTokenIds (CtcDecoder.swift:292-297)."""
    text = "".join(vocabulary[int(i)] for i in ids if int(i) in vocabulary)
    return text.replace(SENTENCEPIECE_WORD_BOUNDARY, " ").strip(" ")


def ctc_greedy_rows(frames_per_utt, blank_id: int, ctx: L.Context | None = None, return_frame_ids: bool = False):
    """The `[[Float]]` overload of ctcGreedyDecode (CtcDecoder.swift:15-36) for a batch of utterances, each a sequence of frames of ANY
    lengths (empty frames allowed): frame[0] seeds the scan (:25), so NaN in column 0 -> index 0; empty frames never touch `prev` (:23).
    -> list of collapsed id arrays (and optionally a list of per-frame ids, -1 for empty frames)."""
    ctx = ctx or L.default_context()
    utts = [list(u) for u in frames_per_utt]
    B = len(utts)
    if B == 0:
        return ([], []) if return_frame_ids else []
    rows = [np.asarray(f, np.float32).reshape(-1) for u in utts for f in u]
    R = len(rows)
    offs = np.zeros(R + 1, np.int64)
    if R:
        offs[1:] = np.cumsum([r.size for r in rows])
    flat = np.concatenate(rows).astype(np.float32, copy=False) if R and offs[-1] else np.zeros(1, np.float32)
    utt_rows = np.zeros(B + 1, np.int64)
    utt_rows[1:] = np.cumsum([len(u) for u in utts])
    tok = np.zeros(max(R, 1), np.int32)
    fids = np.zeros(max(R, 1), np.int32) if return_frame_ids else None
    lens = np.zeros(B, np.int32)
    ctx.check(L.lib().fa_ctc_greedy_rows(ctx.handle, flat.ctypes.data, offs.ctypes.data, R, utt_rows.ctypes.data, B, blank_id,
                                         None if fids is None else fids.ctypes.data, tok.ctypes.data, lens.ctypes.data), "fa_ctc_greedy_rows")
    out = [tok[utt_rows[u]:utt_rows[u] + lens[u]].copy() for u in range(B)]
    if return_frame_ids:
        return out, [fids[utt_rows[u]:utt_rows[u + 1]].copy() for u in range(B)]
    return out


def ctc_greedy_decode(log_probs, vocabulary: dict[int, str], blank_id: int = 1024, ctx: L.Context | None = None) -> str:
    """ctcGreedyDecode(logProbs:vocabulary:blankId:): a list / tuple of frames is the `[[Float]]` overload (:15-36: frame[0] seed, per-frame
    length, empty frames skipped); an array [T, V] / [1, T, V] is the MLMultiArray overload (:45-70: -inf seed, NaN never wins)."""
    if isinstance(log_probs, (list, tuple)):
        if len(log_probs) == 0:
            return ""
        ids = ctc_greedy_rows([log_probs], blank_id, ctx=ctx)[0]
        return decode_ctc_token_ids(ids, vocabulary)
    x = np.asarray(log_probs)
    if x.ndim == 3:
        x = x[0]
    if x.shape[0] == 0:
        return ""
    ids = ctc_greedy_ids_batch(x, blank_id, ctx=ctx)[0]
    return decode_ctc_token_ids(ids, vocabulary)


def ctc_log_probs_dev(ctx, d_logits, temperature: float = 1.0, blank_bias: float = 0.0, blank_id: int = -1, d_out=None,
                      order: bool = True):
    """makeLogProbs (CtcKeywordSpotter+Inference.swift:350-405) for a batch: d_logits torch CUDA [B, T, V] fp32/fp16
    (last dim contiguous) -> float32 CUDA tensor [B, T, V].  Enqueues on ctx.stream."""
    import torch
    B, T, V = d_logits.shape
    assert d_logits.stride(2) == 1
    if d_out is None:
        d_out = torch.empty((B, T, V), dtype=torch.float32, device=d_logits.device)
    dt = L.DTYPE_F16 if d_logits.dtype == torch.float16 else L.DTYPE_F32
    with ctx.torch_ordered(order):
        ctx.check(L.lib().fa_ctc_log_softmax_batch_dev(ctx.handle, C.c_void_p(d_logits.data_ptr()), dt, B, T, V, d_logits.stride(1),
                                                       d_logits.stride(0), float(temperature), float(blank_bias), int(blank_id),
                                                       C.c_void_p(d_out.data_ptr())), "fa_ctc_log_softmax_batch_dev")
    return d_out
