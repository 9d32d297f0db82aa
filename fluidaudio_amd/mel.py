"""Host-side mirror of ``AudioMelSpectrogram``
(reference: Sources/FluidAudio/Shared/AudioMelSpectrogram.swift) over the HIP C ABI.

Same constructor parameters (:59-70), same method names and return tuples
(``computeFlat`` :185-187 -> ``compute_flat`` etc.); the arithmetic runs in
fluidaudio_amd/csrc/mel.hip.  ``MelPlan`` is the batched, device-resident entry the
reference does not have (it processes one utterance per call).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class MelPlan:
    """Fixed batch geometry -> pure kernel launches (fa_mel_plan_*)."""

    def __init__(self, ctx: L.Context, cfg: L.MelConfig, offsets, expected_frames=None, frame_stride: int = 0):
        self.ctx, self.cfg = ctx, cfg
        self.offsets = np.ascontiguousarray(offsets, np.int64)
        self.batch = self.offsets.size - 1
        exp = None if expected_frames is None else np.ascontiguousarray(expected_frames, np.int32)
        self._h = C.c_void_p()
        ctx.check(L.lib().fa_mel_plan_create(ctx.handle, C.byref(cfg), self.offsets.ctypes.data, self.batch,
                                             None if exp is None else exp.ctypes.data, frame_stride, C.byref(self._h)),
                  "fa_mel_plan_create")
        self.utt_stride = L.lib().fa_mel_plan_utt_stride(self._h)
        self.frame_stride = L.lib().fa_mel_plan_frame_stride(self._h)
        self.total_frames = L.lib().fa_mel_plan_total_frames(self._h)

    def out_shape(self):
        if self.cfg.layout == L.MEL_LAYOUT_MEL_MAJOR:
            return (self.batch, self.cfg.n_mels, self.frame_stride)
        return (self.batch, self.frame_stride, self.cfg.n_mels)

    def execute(self, d_pcm, d_mel, d_lengths=None, d_last=None, order: bool = True):
        """All arguments are torch CUDA tensors (float32 pcm/mel/last, int32 lengths).  Enqueues on ctx.stream, ordered
        after torch's current stream and before its later work (Context.torch_ordered); `order=False` skips that."""
        with self.ctx.torch_ordered(order):
            self.ctx.check(L.lib().fa_mel_execute_dev(self._h, _ptr(d_pcm), _ptr(d_last), _ptr(d_mel), _ptr(d_lengths)),
                           "fa_mel_execute_dev")

    def close(self):
        if self._h:
            L.lib().fa_mel_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AudioMelSpectrogram:
    """Drop-in for the reference class of the same name (one instance per thread, like the reference :48-57)."""

    def __init__(self, sample_rate: int = 16000, n_mels: int = 128, n_fft: int = 512, hop_length: int = 160,
                 win_length: int = 400, preemph: float = 0.97, pad_to: int = 0, log_floor: float = 2.0 ** -24,
                 log_floor_mode: str = "additive", window_periodic: bool = False, ctx: L.Context | None = None):
        self.ctx = ctx or L.default_context()
        self.n_fft, self.n_mels = n_fft, n_mels
        self._base = dict(sample_rate=sample_rate, n_mels=n_mels, n_fft=n_fft, hop=hop_length, win=win_length,
                          preemph=preemph, pad_to=pad_to, log_floor=log_floor,
                          floor_mode=L.MEL_FLOOR_CLAMPED if log_floor_mode == "clamped" else L.MEL_FLOOR_ADDITIVE,
                          window_periodic=int(window_periodic))

    def config(self, padding_mode=L.MEL_PAD_CENTER, layout=L.MEL_LAYOUT_MEL_MAJOR) -> L.MelConfig:
        return L.MelConfig(padding_mode=padding_mode, layout=layout, power=2.0, **self._base)

    # -- single-utterance entries with the reference's return tuples ---------------------------
    def _run(self, audio, last, cfg, expected):
        a = np.ascontiguousarray(audio, np.float32).reshape(-1)
        offs = np.array([0, a.size], np.int64)
        T = L.lib().fa_mel_num_frames(C.byref(cfg), a.size)
        if expected is not None and a.size > 0:
            T = max(int(expected), 0)
        if T <= 0:  # guard (:199-201, :349-351)
            return np.zeros(self.n_mels, np.float32), 0, 1
        tpad = L.lib().fa_mel_padded_frames(C.byref(cfg), T)
        out = np.zeros(self.n_mels * tpad, np.float32)
        lens = np.zeros(1, np.int32)
        lastv = np.array([last], np.float32)
        exp = None if expected is None else np.array([expected], np.int32)
        self.ctx.check(L.lib().fa_mel_batch(self.ctx.handle, C.byref(cfg), a.ctypes.data, offs.ctypes.data, 1,
                                            lastv.ctypes.data, None if exp is None else exp.ctypes.data, tpad,
                                            out.ctypes.data, lens.ctypes.data), "fa_mel_batch")
        return out, int(lens[0]), tpad

    def compute_flat(self, audio, last_audio_sample: float = 0.0):
        """computeFlat (:185-292) -> (mel flat [n_mels * numFrames], melLength, numFrames)."""
        return self._run(audio, last_audio_sample, self.config(L.MEL_PAD_CENTER, L.MEL_LAYOUT_MEL_MAJOR), None)

    def compute_flat_transposed(self, audio, last_audio_sample: float = 0.0, padding_mode: str = "center",
                                expected_frame_count: int | None = None):
        """computeFlatTransposed (:325-456) -> (mel flat [numFrames * n_mels], melLength, numFrames)."""
        pm = L.MEL_PAD_PREPADDED if padding_mode in ("prePadded", "prepadded") else L.MEL_PAD_CENTER
        return self._run(audio, last_audio_sample, self.config(pm, L.MEL_LAYOUT_FRAME_MAJOR), expected_frame_count)

    def compute(self, audio):
        """compute (:132-178) -> (mel [1, n_mels, T], melLength)."""
        mel, ml, nf = self._run(audio, 0.0, self.config(L.MEL_PAD_LEGACY, L.MEL_LAYOUT_MEL_MAJOR), None)
        if ml == 0:
            return np.zeros((0, 0, 0), np.float32), 0
        return mel.reshape(1, self.n_mels, nf)[:, :, :ml].copy(), ml

    def get_filterbank(self) -> np.ndarray:
        cfg = self.config()
        out = np.zeros((self.n_mels, self.n_fft // 2 + 1), np.float32)
        self.ctx.check(L.lib().fa_mel_filterbank(C.byref(cfg), out.ctypes.data), "fa_mel_filterbank")
        return out

    def get_hann_window(self) -> np.ndarray:
        cfg = self.config()
        out = np.zeros(cfg.win, np.float32)
        self.ctx.check(L.lib().fa_mel_hann_window(C.byref(cfg), out.ctypes.data), "fa_mel_hann_window")
        return out

    # -- batched, device-resident (no reference counterpart) ------------------------------------
    def plan(self, offsets, layout="mel_major", padding_mode="center", expected_frames=None, frame_stride=0) -> MelPlan:
        pm = {"center": L.MEL_PAD_CENTER, "prepadded": L.MEL_PAD_PREPADDED, "prePadded": L.MEL_PAD_PREPADDED,
              "legacy": L.MEL_PAD_LEGACY}[padding_mode]
        lay = L.MEL_LAYOUT_MEL_MAJOR if layout == "mel_major" else L.MEL_LAYOUT_FRAME_MAJOR
        return MelPlan(self.ctx, self.config(pm, lay), offsets, expected_frames, frame_stride)


class UnifiedMelExtractor:
    """Mirror of ``UnifiedMelExtractor`` (reference: Sources/FluidAudio/ASR/Parakeet/Unified/UnifiedMelExtractor.swift:26-113):
    NeMo-config log-mel of one zero-padded encoder window + per-feature normalisation over the valid frames, returned
    as ``[1, n_mels, totalFrames]`` + the valid frame count.  ``features_batch`` is the device-resident batched form
    (B windows of equal length in one launch pair) the reference does not have."""

    hop_length = 160

    def __init__(self, window_samples: int, n_mels: int = 128, ctx: L.Context | None = None):
        self.window_samples, self.n_mels = window_samples, n_mels
        self.total_frames = window_samples // self.hop_length + 1          # :29
        self.mel = AudioMelSpectrogram(sample_rate=16000, n_mels=n_mels, n_fft=512, hop_length=160, win_length=400,
                                       preemph=0.97, pad_to=0, window_periodic=False, ctx=ctx)

    def features_batch(self, d_windows, valid_counts):
        """d_windows: torch CUDA float32 [B, window_samples]; valid_counts: int sequence.  Returns (mel CUDA tensor
        [B, n_mels, totalFrames], valid frames int32 numpy [B])."""
        import torch
        B = d_windows.shape[0]
        offs = np.arange(B + 1, dtype=np.int64) * self.window_samples
        plan = self.mel.plan(offs, layout="mel_major", expected_frames=np.full(B, self.total_frames, np.int32),
                             frame_stride=self.total_frames)
        d_mel = torch.empty((B, self.n_mels, self.total_frames), dtype=torch.float32, device=d_windows.device)
        d_len = torch.empty(B, dtype=torch.int32, device=d_windows.device)
        valid = np.minimum(np.asarray(valid_counts, np.int64) // self.hop_length, self.total_frames).astype(np.int32)   # :66
        d_valid = torch.from_numpy(valid).to(d_windows.device)
        d_flat = d_windows.contiguous().view(-1)
        ctx = self.mel.ctx
        with ctx.torch_ordered():   # the copies above run on torch's stream, the two kernels on the context's
            plan.execute(d_flat, d_mel, d_len, order=False)
            ctx.check(L.lib().fa_mel_normalize_per_feature_dev(ctx.handle, _ptr(d_mel), B, self.n_mels, self.total_frames,
                                                               self.total_frames, _ptr(d_valid)), "fa_mel_normalize_per_feature_dev")
        ctx.synchronize()
        plan.close()
        return d_mel, valid

    def features(self, window, valid_count: int):
        """features(window:validCount:) (:52-90) -> (mel [1, n_mels, totalFrames] float32 numpy, valid frames)."""
        import torch
        w = np.ascontiguousarray(window, np.float32).reshape(1, -1)
        assert w.shape[1] == self.window_samples
        d_mel, valid = self.features_batch(torch.from_numpy(w).cuda(self.mel.ctx.device), [valid_count])
        return d_mel.cpu().numpy(), int(valid[0])


class LuxTtsMelExtractor:
    """Mirror of ``LuxTtsMelExtractor`` (reference: Sources/FluidAudio/TTS/LuxTts/LuxTtsMelExtractor.swift:15-189), the
    torchaudio-flavoured front end (24 kHz, n_fft 1024, hop 256, 100 mels, periodic Hann of n_fft, reflect padding,
    magnitude spectrum, HTK mel scale without normalisation, log(max(v, 1e-7)), lhotse frame count with the last frame
    replicated).  It is the reference's only mel path with a golden vector in its tests; here it runs through the same
    C ABI (fa_mel_batch with the extension fields of fa_mel_config) on mel_generic_kernel."""

    n_fft, hop, n_mels, sample_rate, log_mel_floor = 1024, 256, 100, 24000, 1e-7   # LuxTtsConstants

    def __init__(self, ctx: L.Context | None = None):
        self.ctx = ctx or L.default_context()

    def frame_count(self, sample_count: int) -> int:
        """frameCount(sampleCount:) (:45-47): lhotse compute_num_frames."""
        return (sample_count + self.hop // 2) // self.hop

    def config(self) -> L.MelConfig:
        return L.MelConfig(sample_rate=self.sample_rate, n_mels=self.n_mels, n_fft=self.n_fft, hop=self.hop, win=self.n_fft,
                           preemph=0.0, pad_to=0, log_floor=self.log_mel_floor, floor_mode=L.MEL_FLOOR_CLAMPED,
                           window_periodic=1, padding_mode=L.MEL_PAD_CENTER, layout=L.MEL_LAYOUT_FRAME_MAJOR, power=1.0,
                           center_pad=L.MEL_CENTER_REFLECT, mel_scale=L.MEL_SCALE_HTK_NONORM, tail_mode=L.MEL_TAIL_REPLICATE)

    def extract(self, audio) -> np.ndarray:
        """extract(audio:) (:52-132) -> [T, n_mels] log-mel (unscaled), T = frame_count(len(audio)); [] for empty input."""
        a = np.ascontiguousarray(audio, np.float32).reshape(-1)
        T = self.frame_count(a.size)
        if a.size == 0 or T <= 0:
            return np.zeros((0, self.n_mels), np.float32)
        cfg = self.config()
        out = np.zeros((T, self.n_mels), np.float32)
        lens = np.zeros(1, np.int32)
        offs = np.array([0, a.size], np.int64)
        exp = np.array([T], np.int32)
        self.ctx.check(L.lib().fa_mel_batch(self.ctx.handle, C.byref(cfg), a.ctypes.data, offs.ctypes.data, 1, None, exp.ctypes.data, T,
                                            out.ctypes.data, lens.ctypes.data), "fa_mel_batch")
        assert int(lens[0]) == T
        return out
