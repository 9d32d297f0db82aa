"""Host-side mirror of the TDT decoder's navigation helpers and greedy control loop (reference:
Sources/FluidAudio/ASR/Parakeet/SlidingWindow/TDT/Decoder/: TdtFrameNavigation.swift:20-105, TdtDurationMapping.swift:17-31,
TdtConfig.swift:13-26, TdtDecoderV3.swift:103-607) over the HIP C ABI (csrc/tdt.hip).

The reference's decoder LSTM and joint network are CoreML bundles that are not part of its source tree, so
``decode_tables`` replays the control loop over tables of the joint's decisions (token, duration bin, probability)
indexed by (decoder steps taken, encoder frame) — token parity against the real models is unpinned."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

STANDARD_OVERLAP_FRAMES = 25  # ASRConstants.standardOverlapFrames


@dataclass
class TdtConfig:  # TdtConfig.swift:13-26
    blank_id: int = 8192
    max_symbols_per_step: int = 10
    max_tokens_per_chunk: int = 150
    consecutive_blank_limit: int = 5
    duration_bins: tuple = (0, 1, 2, 3, 4)

    def c(self) -> L.TdtConfig:
        c = L.TdtConfig()
        c.blank_id, c.max_symbols_per_step = self.blank_id, self.max_symbols_per_step
        c.max_tokens_per_chunk, c.consecutive_blank_limit = self.max_tokens_per_chunk, self.consecutive_blank_limit
        c.n_duration_bins = len(self.duration_bins)
        for i, v in enumerate(self.duration_bins):
            c.duration_bins[i] = v
        return c


class TdtFrameNavigation:
    @staticmethod
    def calculate_initial_time_indices(time_jump, context_frame_adjustment: int) -> int:
        return L.lib().fa_tdt_initial_time_index(0 if time_jump is None else 1, 0 if time_jump is None else int(time_jump),
                                                 int(context_frame_adjustment))

    @staticmethod
    def initialize_navigation_state(time_indices: int, encoder_sequence_length: int, actual_audio_frames: int):
        e, s, l, a = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        L.lib().fa_tdt_navigation_state(time_indices, encoder_sequence_length, actual_audio_frames, C.byref(e), C.byref(s),
                                        C.byref(l), C.byref(a))
        return e.value, s.value, l.value, bool(a.value)

    @staticmethod
    def calculate_final_time_jump(current_time_indices: int, effective_sequence_length: int, is_last_chunk: bool):
        has = C.c_int32()
        v = L.lib().fa_tdt_final_time_jump(current_time_indices, effective_sequence_length, int(is_last_chunk), C.byref(has))
        return v if has.value else None


class TdtDurationMapping:
    @staticmethod
    def map_duration_bin(bin_index: int, duration_bins) -> int:
        cfg = TdtConfig(duration_bins=tuple(duration_bins)).c()
        out = C.c_int32()
        if L.lib().fa_tdt_map_duration_bin(C.byref(cfg), int(bin_index), C.byref(out)) != L.SUCCESS:
            raise ValueError(f"Duration bin index out of range: {bin_index}")  # ASRError.processingFailed (:19-21)
        return out.value

    @staticmethod
    def clamp_probability(value: float) -> float:
        return float(L.lib().fa_tdt_clamp_probability(float(value)))


def decode_logits(d_logits, vocab_with_blank: int, enc_len, audio_frames=None, t0=None, is_last=None, global_offset=None, emit_after=None,
                  config: TdtConfig | None = None, max_out: int = 256, ctx: L.Context | None = None):
    """Batched greedy walk on joint LOGITS: d_logits torch CUDA tensor [B, U, T, W] (fp32 / fp16, last dim contiguous, W >=
    vocab_with_blank + number of duration bins).  Same result structure as decode_tables."""
    import torch
    ctx = ctx or L.default_context()
    cfg = (config or TdtConfig()).c()
    B, U, T, W = d_logits.shape
    assert d_logits.is_contiguous()
    dev = d_logits.device

    def vec(v):
        return None if v is None else torch.as_tensor(np.asarray(v, np.int32)).to(dev)

    v_enc, v_af, v_t0, v_last, v_go, v_ea = (vec(enc_len), vec(audio_frames), vec(t0), vec(is_last), vec(global_offset),
                                             vec(None if emit_after is None else [-1 if e is None else e for e in emit_after]))
    o_tok, o_time, o_dur = (torch.zeros((B, max_out), dtype=torch.int32, device=dev) for _ in range(3))
    o_conf = torch.zeros((B, max_out), dtype=torch.float32, device=dev)
    o_cnt, o_ft, o_fu, o_st = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(4))
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    dt = L.DTYPE_F16 if d_logits.dtype == torch.float16 else L.DTYPE_F32
    with ctx.torch_ordered():
        ctx.check(L.lib().fa_tdt_greedy_logits_dev(ctx.handle, C.byref(cfg), p(d_logits), dt, B, U, T, int(vocab_with_blank), W, p(v_enc), p(v_af), p(v_t0),
                                                   p(v_last), p(v_go), p(v_ea), max_out, p(o_tok), p(o_time), p(o_dur), p(o_conf), p(o_cnt), p(o_ft),
                                                   p(o_fu), p(o_st)), "fa_tdt_greedy_logits_dev")
    ctx.synchronize()
    tok, tim, dur, conf = o_tok.cpu().numpy(), o_time.cpu().numpy(), o_dur.cpu().numpy(), o_conf.cpu().numpy()
    cnt, ft, fu, st = o_cnt.cpu().numpy(), o_ft.cpu().numpy(), o_fu.cpu().numpy(), o_st.cpu().numpy()
    out = []
    for b in range(B):
        n = min(int(cnt[b]), max_out)
        out.append(dict(status=int(st[b]), tokens=tok[b, :n].copy(), timestamps=tim[b, :n].copy(), durations=dur[b, :n].copy(),
                        confidences=conf[b, :n].copy(), count=int(cnt[b]), final_time=None if ft[b] == -2 ** 31 else int(ft[b]),
                        final_u=int(fu[b])))
    return out


def decode_tables(d_tok, d_bin, d_prob, enc_len, audio_frames=None, t0=None, is_last=None, global_offset=None, emit_after=None,
                  config: TdtConfig | None = None, max_out: int = 256, ctx: L.Context | None = None):
    """Batched greedy walk.  d_tok/d_bin (int32) and d_prob (float32): torch CUDA tensors [B, U, T]; per-chunk int
    sequences for the rest.  Returns a list of dicts (tokens, timestamps, durations, confidences, final_time, final_u, status)."""
    import torch
    ctx = ctx or L.default_context()
    cfg = (config or TdtConfig()).c()
    B, U, T = d_tok.shape
    dev = d_tok.device

    def vec(v):
        return None if v is None else torch.as_tensor(np.asarray(v, np.int32)).to(dev)

    v_enc, v_af, v_t0, v_last, v_go, v_ea = (vec(enc_len), vec(audio_frames), vec(t0), vec(is_last), vec(global_offset),
                                             vec(None if emit_after is None else [-1 if e is None else e for e in emit_after]))
    o_tok, o_time, o_dur = (torch.zeros((B, max_out), dtype=torch.int32, device=dev) for _ in range(3))
    o_conf = torch.zeros((B, max_out), dtype=torch.float32, device=dev)
    o_cnt, o_ft, o_fu, o_st = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(4))
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    c_tok, c_bin, c_prob = d_tok.contiguous(), d_bin.contiguous(), d_prob.contiguous()   # may launch copies on torch's stream
    with ctx.torch_ordered():
        ctx.check(L.lib().fa_tdt_greedy_tables_dev(ctx.handle, C.byref(cfg), p(c_tok), p(c_bin), p(c_prob),
                                                   B, U, T, p(v_enc), p(v_af), p(v_t0), p(v_last), p(v_go), p(v_ea), max_out, p(o_tok),
                                                   p(o_time), p(o_dur), p(o_conf), p(o_cnt), p(o_ft), p(o_fu), p(o_st)), "fa_tdt_greedy_tables_dev")
    ctx.synchronize()
    tok, tim, dur, conf = o_tok.cpu().numpy(), o_time.cpu().numpy(), o_dur.cpu().numpy(), o_conf.cpu().numpy()
    cnt, ft, fu, st = o_cnt.cpu().numpy(), o_ft.cpu().numpy(), o_fu.cpu().numpy(), o_st.cpu().numpy()
    out = []
    for b in range(B):
        n = min(int(cnt[b]), max_out)
        out.append(dict(status=int(st[b]), tokens=tok[b, :n].copy(), timestamps=tim[b, :n].copy(), durations=dur[b, :n].copy(),
                        confidences=conf[b, :n].copy(), count=int(cnt[b]), final_time=None if ft[b] == -2 ** 31 else int(ft[b]),
                        final_u=int(fu[b])))
    return out
