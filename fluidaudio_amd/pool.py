"""Device set for single-process hosts (csrc/pool.hip): one context per listed device, handed to concurrent callers, and the
host-pointer entries sharded across all of them.  The reference has no multi-device notion; the partitioning is SURVEY §8e's
(independent utterances / logit matrices / recordings, nothing exchanged).  Multi-PROCESS sharding (one rank per GPU over
torch.distributed) lives in sharding.py; this is the form a Swift / C host gets through the C ABI."""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import _lib as L


def device_count() -> int:
    n = C.c_int32(0)
    st = L.lib().fa_device_count(C.byref(n))
    if st != L.SUCCESS:
        raise L.FluidAudioHipError(st, "fa_device_count", "no usable MI355X visible")
    return int(n.value)


class Pool:
    """fa_pool.  devices=None: every visible device; a device may be listed more than once (several streams on it)."""

    def __init__(self, devices=None):
        self._h = C.c_void_p()
        if devices is None:
            st = L.lib().fa_pool_create(None, 0, C.byref(self._h))
        else:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            st = L.lib().fa_pool_create(arr, len(devices), C.byref(self._h))
        if st != L.SUCCESS:
            raise L.FluidAudioHipError(st, "fa_pool_create", "no usable MI355X visible" if st == L.RUNTIME_ERROR else "")

    def __len__(self) -> int:
        return int(L.lib().fa_pool_size(self._h))

    def devices(self) -> list[int]:
        return [int(L.lib().fa_ctx_device(L.lib().fa_pool_context(self._h, i))) for i in range(len(self))]

    @contextlib.contextmanager
    def acquire(self):
        """Borrow one context (blocks while all are taken); yields (raw fa_ctx handle, device id)."""
        h = C.c_void_p()
        st = L.lib().fa_pool_acquire(self._h, C.byref(h))
        if st != L.SUCCESS:
            raise L.FluidAudioHipError(st, "fa_pool_acquire")
        try:
            yield h, int(L.lib().fa_ctx_device(h))
        finally:
            L.lib().fa_pool_release(self._h, h)

    def _check(self, st: int, where: str):
        if st != L.SUCCESS:
            detail = "; ".join(filter(None, ((L.lib().fa_ctx_last_error(L.lib().fa_pool_context(self._h, i)) or b"").decode() for i in range(len(self)))))
            raise L.FluidAudioHipError(st, where, detail)

    def mel_batch(self, cfg: L.MelConfig, utterances, last_samples=None, expected_frames=None, frame_stride: int = 0):
        """fa_mel_batch_sharded over host buffers (cfg e.g. AudioMelSpectrogram().config()): returns (mel [B, n_mels, frame_stride] or [B, frame_stride, n_mels], lengths)."""
        utts = [np.ascontiguousarray(u, np.float32).ravel() for u in utterances]
        B = len(utts)
        offsets = np.zeros(B + 1, np.int64)
        offsets[1:] = np.cumsum([u.size for u in utts])
        pcm = np.concatenate(utts) if B and offsets[-1] else np.zeros(1, np.float32)
        fs = int(frame_stride)
        if fs <= 0:
            fs = 1
            for b in range(B):
                T = int(L.lib().fa_mel_num_frames(C.byref(cfg), int(utts[b].size)))
                if expected_frames is not None and utts[b].size > 0:
                    T = max(int(expected_frames[b]), 0)
                fs = max(fs, int(L.lib().fa_mel_padded_frames(C.byref(cfg), T)) if T > 0 else 1)
        shape = (B, cfg.n_mels, fs) if cfg.layout == L.MEL_LAYOUT_MEL_MAJOR else (B, fs, cfg.n_mels)
        mel = np.zeros(shape, np.float32)
        lens = np.zeros(B, np.int32)
        last = None if last_samples is None else np.ascontiguousarray(last_samples, np.float32)
        exp = None if expected_frames is None else np.ascontiguousarray(expected_frames, np.int32)
        st = L.lib().fa_mel_batch_sharded(self._h, C.byref(cfg), pcm.ctypes.data, offsets.ctypes.data, B,
                                          None if last is None else last.ctypes.data, None if exp is None else exp.ctypes.data, fs,
                                          mel.ctypes.data, lens.ctypes.data)
        self._check(st, "fa_mel_batch_sharded")
        return mel, lens

    def ctc_greedy_batch(self, logits, blank_id: int, valid_frames=None, return_frame_ids: bool = False):
        """fa_ctc_greedy_batch_sharded: logits [B, T, V] float32 / float16 host array -> (token_ids [B, T], token_lens [B])."""
        x = np.ascontiguousarray(logits)
        assert x.ndim == 3 and x.dtype in (np.float32, np.float16)
        B, T, V = x.shape
        ids = np.zeros((B, T), np.int32)
        lens = np.zeros(B, np.int32)
        fids = np.zeros((B, T), np.int32) if return_frame_ids else None
        vf = None if valid_frames is None else np.ascontiguousarray(valid_frames, np.int32)
        st = L.lib().fa_ctc_greedy_batch_sharded(self._h, x.ctypes.data, L.DTYPE_F16 if x.dtype == np.float16 else L.DTYPE_F32, B, T, V, V, T * V,
                                                 None if vf is None else vf.ctypes.data, int(blank_id), None if fids is None else fids.ctypes.data,
                                                 ids.ctypes.data, lens.ctypes.data)
        self._check(st, "fa_ctc_greedy_batch_sharded")
        return (ids, lens, fids) if return_frame_ids else (ids, lens)

    def linkage_many(self, problems, mode: int = L.AHC_MODE_AUTO):
        """fa_ahc_linkage_many: recordings dealt across the devices; returns (statuses, [Z_k])."""
        xs = [np.ascontiguousarray(p, np.float64) for p in problems]
        k = len(xs)
        if k == 0:
            return [], []
        d = xs[0].shape[1]
        assert all(x.ndim == 2 and x.shape[1] == d for x in xs)
        zs = [np.zeros((max(x.shape[0] - 1, 0), 4), np.float64) for x in xs]
        dummy = np.zeros(4)
        zp = (C.c_void_p * k)(*[z.ctypes.data if z.size else dummy.ctypes.data for z in zs])
        dp = (C.c_void_p * k)(*[x.ctypes.data if x.size else dummy.ctypes.data for x in xs])
        ns = (C.c_size_t * k)(*[x.shape[0] for x in xs])
        st = (C.c_int32 * k)()
        L.lib().fa_ahc_linkage_many(self._h, k, dp, ns, d, zp, mode, None, st)
        return [int(v) for v in st], zs

    def close(self):
        if self._h:
            L.lib().fa_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
