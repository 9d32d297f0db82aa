"""CPU ORACLE bindings (test infrastructure, NOT product code).

ctypes wrappers over ``oracle/libfa_oracle.so`` (C restatement, see fa_oracle.h) and
``oracle/_ref/libfastcluster_ref.so`` (the reference's own FastClusterWrapper C++ built
from /root/reference by oracle/Makefile).  Only tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg may import this package.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfa_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libfastcluster_ref.so")


def build(force: bool = False) -> None:
    """Compile the C restatement and (when /root/reference exists) oracle/_ref."""
    stale = os.path.exists(_LIB_PATH) and os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("fa_oracle.c", "fa_oracle.h"))
    if force or stale or not os.path.exists(_LIB_PATH) or not os.path.exists(_REF_PATH):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


class _MelCfg(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int), ("n_mels", C.c_int), ("n_fft", C.c_int), ("hop", C.c_int),
        ("win", C.c_int), ("preemph", C.c_float), ("pad_to", C.c_int), ("log_floor", C.c_float),
        ("floor_clamped", C.c_int), ("window_periodic", C.c_int),
    ]


@dataclass
class MelConfig:
    sample_rate: int = 16000
    n_mels: int = 128
    n_fft: int = 512
    hop: int = 160
    win: int = 400
    preemph: float = 0.97
    pad_to: int = 0
    log_floor: float = 2.0 ** -24
    floor_clamped: bool = False
    window_periodic: bool = False

    def c(self) -> _MelCfg:
        return _MelCfg(self.sample_rate, self.n_mels, self.n_fft, self.hop, self.win, self.preemph,
                       self.pad_to, self.log_floor, int(self.floor_clamped), int(self.window_periodic))


_lib = None
_ref = None
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
LINKAGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.fa_oracle_hann.argtypes = [C.c_int, C.c_int, _f32p]
        L.fa_oracle_slaney_filterbank.argtypes = [C.c_int, C.c_int, C.c_int, _f32p]
        L.fa_oracle_mel_frames_center.argtypes = [C.POINTER(_MelCfg), C.c_long]
        L.fa_oracle_mel_frames_prepadded.argtypes = [C.POINTER(_MelCfg), C.c_long]
        L.fa_oracle_mel_padded_frames.argtypes = [C.POINTER(_MelCfg), C.c_int]
        L.fa_oracle_mel_flat.argtypes = [C.POINTER(_MelCfg), _f32p, C.c_long, C.c_float, _f32p,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.fa_oracle_mel_flat_transposed.argtypes = [C.POINTER(_MelCfg), _f32p, C.c_long, C.c_float, C.c_int,
                                                    C.c_int, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.fa_oracle_mel_legacy.argtypes = [C.POINTER(_MelCfg), _f32p, C.c_long, _f32p]
        L.fa_oracle_logmel_generic.argtypes = [_f32p, C.c_long, C.c_int, C.c_int, _f32p, _f32p, C.c_int,
                                               C.c_int, C.c_float, C.c_int, _f32p]
        L.fa_oracle_fft_f32.argtypes = [C.c_int, _f32p, _f32p]
        L.fa_oracle_argmax_rows.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long, _i32p]
        L.fa_oracle_ctc_collapse.argtypes = [_i32p, C.c_long, C.c_int32, _i32p]
        L.fa_oracle_ctc_collapse.restype = C.c_long
        L.fa_oracle_ctc_greedy.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int32, _i32p]
        L.fa_oracle_ctc_greedy.restype = C.c_long
        L.fa_oracle_ctc_greedy_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int32, C.c_void_p, _i32p]
        L.fa_oracle_ctc_greedy_rows.restype = C.c_long
        L.fa_oracle_ahc_normalize.argtypes = [_f64p, C.c_long, C.c_long, _f64p]
        L.fa_oracle_ahc_clamp_threshold.argtypes = [C.c_double]
        L.fa_oracle_ahc_clamp_threshold.restype = C.c_double
        L.fa_oracle_ahc_cut.argtypes = [_f64p, C.c_long, C.c_double, _i32p]
        L.fa_oracle_ahc_cluster.argtypes = [C.c_void_p, _f64p, C.c_long, C.c_long, C.c_double, _i32p]
        L.fa_oracle_linkage_naive.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        L.fa_oracle_vbx_refine.argtypes = [_f64p, C.c_long, C.c_long, _i32p, _f64p, C.c_int, C.c_double,
                                           C.c_double, C.c_double, _f64p, _f64p, _i32p, _f64p,
                                           C.POINTER(C.c_long)]
        L.fa_oracle_weighted_centroids.argtypes = [_f64p, C.c_long, C.c_long, _f64p, _f64p, C.c_long, _f64p, _i32p]
        L.fa_oracle_weighted_centroids.restype = C.c_long
        L.fa_oracle_assign_cosine.argtypes = [_f64p, C.c_long, C.c_long, _f64p, C.c_long, _i32p]
        L.fa_oracle_log_softmax_row.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_int, _f32p]
        L.fa_oracle_seeded_rng_next.argtypes = [C.POINTER(C.c_uint64)]
        L.fa_oracle_seeded_rng_next.restype = C.c_uint64
        L.fa_oracle_rng_upper_bound.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.fa_oracle_rng_upper_bound.restype = C.c_uint64
        L.fa_oracle_shuffle_indices.argtypes = [C.POINTER(C.c_uint64), C.c_long, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
        L.fa_oracle_shuffle_indices.restype = None
        L.fa_oracle_kmeans.argtypes = [_f64p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_uint64, _i32p, _f64p,
                                       C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.fa_oracle_kmeans_ninit.argtypes = [_f64p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_uint64, _i32p, _f64p,
                                             C.POINTER(C.c_long), C.POINTER(C.c_long), _f64p]
        L.fa_oracle_speaker_constraints.argtypes = [C.c_long, C.c_int, C.c_long, C.c_int, C.c_long, C.c_int, C.c_long,
                                                    C.POINTER(C.c_long * 3)]
        L.fa_oracle_speaker_constraints.restype = None
        L.fa_oracle_log_softmax_row.restype = None
        L.fa_oracle_tdt_initial_time_index.argtypes = [C.c_int, C.c_int, C.c_int]
        L.fa_oracle_tdt_clamp_probability.argtypes = [C.c_float]
        L.fa_oracle_tdt_clamp_probability.restype = C.c_float
        L.fa_oracle_tdt_last_joint_calls.restype = C.c_long
        L.fa_oracle_tdt_greedy.argtypes = [_i32p, _i32p, _f32p] + [C.c_int] * 12 + [_i32p, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f32p,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.fa_oracle_hungarian_solve.argtypes = [np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS"), C.c_int, _i32p]
        L.fa_oracle_hungarian_solve.restype = None
        L.fa_oracle_max_score_assignment.argtypes = [_f64p, C.c_int, C.c_int, _i32p]
        L.fa_oracle_max_score_assignment.restype = None
        L.fa_oracle_constrained_assign.argtypes = [_f64p, C.c_long, C.c_int, _i32p, _i32p]
        L.fa_oracle_constrained_assign.restype = None
        L.fa_oracle_centroid_scores.argtypes = [_f64p, C.c_long, C.c_long, _f64p, C.c_long, _f64p]
        L.fa_oracle_centroid_scores.restype = None
        L.fa_oracle_normalize_per_feature.argtypes = [_f32p, C.c_int, C.c_int, C.c_int]
        L.fa_oracle_normalize_per_feature.restype = None
        L.fa_oracle_resample_linear_frames.argtypes = [C.c_long, C.c_double, C.c_double]
        L.fa_oracle_resample_linear_frames.restype = C.c_long
        L.fa_oracle_resample_linear.argtypes = [_f32p, C.c_int, C.c_long, C.c_double, C.c_double, _f32p]
        L.fa_oracle_resample_linear.restype = C.c_long
    return _lib


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref() -> C.CDLL:
    """The reference's own fastcluster C ABI, built from /root/reference (oracle/_ref)."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_PATH):
            build()
        _ref = C.CDLL(_REF_PATH)
        _ref.fastcluster_compute_centroid_linkage.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t,
                                                              C.c_void_p, C.c_size_t]
        _ref.fastcluster_compute_centroid_linkage.restype = C.c_int
    return _ref


# ------------------------------------------------------------------ mel
def hann(win: int, periodic: bool = False) -> np.ndarray:
    out = np.zeros(win, np.float32)
    lib().fa_oracle_hann(win, int(periodic), out)
    return out


def slaney_filterbank(n_fft=512, n_mels=128, sample_rate=16000) -> np.ndarray:
    out = np.zeros((n_mels, n_fft // 2 + 1), np.float32)
    lib().fa_oracle_slaney_filterbank(n_fft, n_mels, sample_rate, out)
    return out


def mel_frames(cfg: MelConfig, n: int, prepadded: bool = False) -> int:
    c = cfg.c()
    f = lib().fa_oracle_mel_frames_prepadded if prepadded else lib().fa_oracle_mel_frames_center
    return f(C.byref(c), n)


def mel_flat(audio, cfg: MelConfig = MelConfig(), last: float = 0.0):
    """computeFlat -> (mel [n_mels, Tpad], melLength, numFrames)."""
    a = np.ascontiguousarray(audio, np.float32)
    c = cfg.c()
    T = lib().fa_oracle_mel_frames_center(C.byref(c), a.size)
    tpad = max(1, lib().fa_oracle_mel_padded_frames(C.byref(c), T))
    out = np.zeros((cfg.n_mels, tpad), np.float32)
    ml, nf = C.c_int(), C.c_int()
    rc = lib().fa_oracle_mel_flat(C.byref(c), a if a.size else np.zeros(1, np.float32), a.size, last, out,
                                  C.byref(ml), C.byref(nf))
    assert rc == 0
    return out, ml.value, nf.value


def mel_flat_transposed(audio, cfg: MelConfig = MelConfig(), last: float = 0.0, prepadded: bool = False,
                        expected_frames: int | None = None):
    """computeFlatTransposed -> (mel [Tpad, n_mels], melLength, numFrames)."""
    a = np.ascontiguousarray(audio, np.float32)
    c = cfg.c()
    T = mel_frames(cfg, a.size, prepadded)
    if expected_frames is not None and a.size > 0:
        T = expected_frames
    tpad = max(1, lib().fa_oracle_mel_padded_frames(C.byref(c), T)) if T > 0 else 1
    out = np.zeros((tpad, cfg.n_mels), np.float32)
    ml, nf = C.c_int(), C.c_int()
    rc = lib().fa_oracle_mel_flat_transposed(C.byref(c), a if a.size else np.zeros(1, np.float32), a.size, last,
                                             int(prepadded), -1 if expected_frames is None else expected_frames,
                                             out, C.byref(ml), C.byref(nf))
    assert rc == 0
    return out, ml.value, nf.value


def mel_legacy(audio, cfg: MelConfig = MelConfig()):
    a = np.ascontiguousarray(audio, np.float32)
    T = 1 + int((a.size - cfg.win) / cfg.hop)  # truncating division like Swift
    if T <= 0:
        return np.zeros((cfg.n_mels, 0), np.float32), 0
    out = np.zeros((cfg.n_mels, T), np.float32)
    c = cfg.c()
    got = lib().fa_oracle_mel_legacy(C.byref(c), a, a.size, out)
    assert got == T
    return out, T


def logmel_generic(audio, n_fft, hop, window, fb, power, floor_v, frames):
    a = np.ascontiguousarray(audio, np.float32)
    fb = np.ascontiguousarray(fb, np.float32)
    out = np.zeros((frames, fb.shape[0]), np.float32)
    lib().fa_oracle_logmel_generic(a, a.size, n_fft, hop, np.ascontiguousarray(window, np.float32), fb,
                                   fb.shape[0], power, floor_v, frames, out)
    return out


# ------------------------------------------------------------------ argmax / CTC
def _logits_args(logits: np.ndarray, row_stride):
    assert logits.dtype in (np.float32, np.float16) and logits.ndim == 2
    x = np.ascontiguousarray(logits)
    T, W = x.shape
    return x, int(x.dtype == np.float16), T, (W if row_stride is None else row_stride)


def argmax_rows(logits: np.ndarray, vocab: int | None = None, frames: int | None = None) -> np.ndarray:
    """logits [T, row_stride]; scans the first `vocab` columns of the first `frames` rows."""
    x, f16, T, stride = _logits_args(logits, None)
    V = stride if vocab is None else vocab
    F = T if frames is None else frames
    ids = np.zeros(max(F, 1), np.int32)
    lib().fa_oracle_argmax_rows(x.ctypes.data, f16, F, V, stride, ids)
    return ids[:F]


def ctc_collapse(ids, blank_id: int) -> np.ndarray:
    ids = np.ascontiguousarray(ids, np.int32)
    out = np.zeros(max(ids.size, 1), np.int32)
    n = lib().fa_oracle_ctc_collapse(ids if ids.size else np.zeros(1, np.int32), ids.size, blank_id, out)
    return out[:n]


def ctc_greedy(logits: np.ndarray, blank_id: int, vocab: int | None = None, frames: int | None = None):
    x, f16, T, stride = _logits_args(logits, None)
    V = stride if vocab is None else vocab
    F = T if frames is None else frames
    out = np.zeros(max(F, 1), np.int32)
    n = lib().fa_oracle_ctc_greedy(x.ctypes.data, f16, F, V, stride, blank_id, out)
    return out[:n]


def ctc_greedy_rows(frames, blank_id: int, return_frame_ids: bool = False):
    """ctcGreedyDecode(logProbs: [[Float]]) (CtcDecoder.swift:15-36) on a sequence of frames of any lengths (C restatement)."""
    rows = [np.asarray(f, np.float32).reshape(-1) for f in frames]
    R = len(rows)
    offs = np.zeros(R + 1, np.int64)
    if R:
        offs[1:] = np.cumsum([r.size for r in rows])
    flat = np.ascontiguousarray(np.concatenate(rows), np.float32) if R and offs[-1] else np.zeros(1, np.float32)
    out = np.zeros(max(R, 1), np.int32)
    fids = np.zeros(max(R, 1), np.int32)
    n = lib().fa_oracle_ctc_greedy_rows(flat.ctypes.data, offs.ctypes.data, R, blank_id, fids.ctypes.data, out)
    return (out[:n], fids[:R]) if return_frame_ids else out[:n]


def ctc_greedy_rows_py(frames, blank_id: int):
    """The same overload as a literal pure-Python transcription of :20-34 (second restatement, small cases only)."""
    ids, prev = [], -1
    for frame in frames:
        frame = [np.float32(v) for v in frame]
        if len(frame) == 0:
            continue
        best_idx, best_val = 0, frame[0]
        for v in range(1, len(frame)):
            if frame[v] > best_val:
                best_val, best_idx = frame[v], v
        if best_idx != blank_id and best_idx != prev:
            ids.append(best_idx)
        prev = best_idx
    return np.asarray(ids, np.int32)


# ------------------------------------------------------------------ AHC
def ahc_normalize(x) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros_like(x)
    if x.size:
        lib().fa_oracle_ahc_normalize(x, x.shape[0], x.shape[1], out)
    return out


def ahc_cut(z, n: int, threshold: float) -> np.ndarray:
    z = np.ascontiguousarray(z, np.float64).reshape(-1)
    labels = np.zeros(max(n, 1), np.int32)
    lib().fa_oracle_ahc_cut(z if z.size else np.zeros(4), n, lib().fa_oracle_ahc_clamp_threshold(threshold), labels)
    return labels[:n]


def linkage_ref(x) -> tuple[int, np.ndarray]:
    """fastcluster_compute_centroid_linkage from the reference's own C++ (oracle/_ref)."""
    x = np.ascontiguousarray(x, np.float64)
    n, d = x.shape
    z = np.zeros((max(n - 1, 0), 4), np.float64)
    st = ref().fastcluster_compute_centroid_linkage(x.ctypes.data, n, d, z.ctypes.data, z.size)
    return st, z


def linkage_naive(x) -> tuple[int, np.ndarray]:
    x = np.ascontiguousarray(x, np.float64)
    n, d = x.shape
    z = np.zeros((max(n - 1, 0), 4), np.float64)
    st = lib().fa_oracle_linkage_naive(x.ctypes.data, n, d, z.ctypes.data, z.size)
    return st, z


def ahc_cluster(x, threshold: float, linkage=None) -> np.ndarray:
    """AHCClustering.cluster; `linkage` = ctypes function with the reference C ABI
    (default: the reference's own build in oracle/_ref)."""
    x = np.asarray(x, np.float64)
    if x.ndim != 2:
        x = x.reshape(len(x), -1)
    n, d = x.shape
    if n == 0:
        return np.zeros(0, np.int32)
    fn = linkage if linkage is not None else ref().fastcluster_compute_centroid_linkage
    labels = np.zeros(n, np.int32)
    xx = np.ascontiguousarray(x) if d > 0 else np.zeros((n, 1))
    lib().fa_oracle_ahc_cluster(C.cast(fn, C.c_void_p), xx, n, d, threshold, labels)
    return labels


# ------------------------------------------------------------------ VBx / post-VBx
def vbx_refine(rho, initial, phi, max_iter=20, epsilon=1e-4, Fa=0.07, Fb=0.8):
    rho = np.ascontiguousarray(rho, np.float64)
    T, D = rho.shape
    initial = np.ascontiguousarray(initial, np.int32)
    S = len(np.unique(initial))
    gamma = np.zeros((T, S), np.float64)
    pi = np.zeros(S, np.float64)
    hard = np.zeros(T, np.int32)
    elbos = np.zeros(max(max_iter, 1), np.float64)
    s_out = C.c_long()
    it = lib().fa_oracle_vbx_refine(rho, T, D, initial, np.ascontiguousarray(phi, np.float64), max_iter,
                                    epsilon, Fa, Fb, gamma, pi, hard, elbos, C.byref(s_out))
    assert s_out.value == S
    return gamma, pi, hard, elbos[:it]


def vbx_refine_degraded(initial):
    """What VBxClustering.refine returns when runVBx throws (VBxClustering.swift:136-146): gamma = initialGamma — the plain one-hot of
    max(0, min(cluster, S - 1)) (:100-104), pi = 1/S (:139), no ELBOs (:140), hardClusters = first-max argmax of gamma (:144-146)."""
    initial = np.ascontiguousarray(initial, np.int32)
    T = initial.size
    S = max(1, len(np.unique(initial)))                                            # :78
    gamma = np.zeros((T, S), np.float64)
    for i, c in enumerate(initial.tolist()):
        gamma[i, max(0, min(c, S - 1))] = 1.0
    pi = np.full(S, 1.0 / S)
    hard = np.argmax(gamma, axis=1).astype(np.int32) if T else np.zeros(0, np.int32)
    return gamma, pi, hard, np.zeros(0)


def weighted_centroids(emb, gamma, pi):
    emb = np.ascontiguousarray(emb, np.float64)
    gamma = np.ascontiguousarray(gamma, np.float64)
    pi = np.ascontiguousarray(pi, np.float64)
    n, d = emb.shape
    S = pi.size
    cent = np.zeros((S, d), np.float64)
    mp = np.zeros(S, np.int32)
    K = lib().fa_oracle_weighted_centroids(emb, n, d, gamma, pi, S, cent, mp)
    return cent[:K], mp


def assign_cosine(emb, centroids):
    emb = np.ascontiguousarray(emb, np.float64)
    centroids = np.ascontiguousarray(centroids, np.float64)
    out = np.zeros(emb.shape[0], np.int32)
    lib().fa_oracle_assign_cosine(emb, emb.shape[0], emb.shape[1], centroids if centroids.size else np.zeros((1, 1)),
                                  centroids.shape[0], out)
    return out


def resample_linear(channel_data, sample_rate: float, target_rate: float = 16000.0) -> np.ndarray:
    """AudioConverter.linearResample (AudioConverter.swift:388-442) on planar [channels, frames] float32."""
    x = np.ascontiguousarray(channel_data, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    ch, frames = x.shape
    n = lib().fa_oracle_resample_linear_frames(frames, float(sample_rate), float(target_rate))
    out = np.zeros(max(n, 1), np.float32)
    got = lib().fa_oracle_resample_linear(x, ch, frames, float(sample_rate), float(target_rate), out)
    return out[:got]


def unified_mel_features(window, valid_count: int, n_mels: int = 128):
    """UnifiedMelExtractor.features (UnifiedMelExtractor.swift:52-90): (mel [n_mels, totalFrames], valid frames)."""
    window = np.ascontiguousarray(window, np.float32)
    total = window.size // 160 + 1
    cfg = MelConfig(n_mels=n_mels)
    flat, _, _ = mel_flat_transposed(window, cfg, 0.0, expected_frames=total)
    flat = np.ascontiguousarray(flat.reshape(total, n_mels), np.float32)
    valid = min(valid_count // 160, total)
    lib().fa_oracle_normalize_per_feature(flat, n_mels, total, valid)
    return np.ascontiguousarray(flat.T), valid


def hungarian_solve(cost, n: int) -> np.ndarray:
    out = np.zeros(max(n, 1), np.int32)
    if n:
        lib().fa_oracle_hungarian_solve(np.ascontiguousarray(cost, np.int64).reshape(-1), n, out)
    return out[:n]


def max_score_assignment(scores) -> np.ndarray:
    rows = len(scores)
    cols = len(scores[0]) if rows else 0
    out = np.zeros(max(rows, 1), np.int32)
    if rows:
        sc = np.ascontiguousarray(scores, np.float64).reshape(rows, cols) if cols else np.zeros((rows, 1))
        lib().fa_oracle_max_score_assignment(sc, rows, cols, out)
    return out[:rows]


def constrained_assign(scores, chunk_indices) -> np.ndarray:
    sc = np.ascontiguousarray(scores, np.float64)
    n = len(chunk_indices)
    K = sc.shape[1] if sc.ndim == 2 else 0
    out = np.zeros(max(n, 1), np.int32)
    if n:
        lib().fa_oracle_constrained_assign(sc if sc.size else np.zeros((1, 1)), n, K, np.ascontiguousarray(chunk_indices, np.int32), out)
    return out[:n]


def centroid_scores(emb, centroids) -> np.ndarray:
    emb = np.ascontiguousarray(emb, np.float64)
    cen = np.ascontiguousarray(centroids, np.float64)
    out = np.zeros((emb.shape[0], cen.shape[0]), np.float64)
    if out.size:
        lib().fa_oracle_centroid_scores(emb, emb.shape[0], emb.shape[1], cen, cen.shape[0], out)
    return out


def cluster_embeddings(embedding256, rho128, chunk_indices, phi, threshold=0.6, Fa=0.07, Fb=0.8, max_iter=20, tol=1e-4,
                       constrained=True, num_speakers=None, min_speakers=None, max_speakers=None, initial=None, vbx_fails=False,
                       ahc_fails=False):
    """CPU restatement of OfflineDiarizerManager.cluster (:270-375) on precomputed embeddings, including the speaker-count
    constraints of VBxClustering.refineWithConstraints (VBxClustering.swift:685-733)."""
    e32 = np.asarray(embedding256, np.float32)
    emb = e32.astype(np.float64)
    ok = np.isfinite(e32).all(axis=1)
    train = np.nonzero(ok)[0] if ok.any() else np.arange(len(e32))
    temb, trho = emb[train], np.ascontiguousarray(rho128, np.float64)[train]
    if initial is None:       # `initial`: AHC labels of the same input computed earlier (the 8 h digest runs two variants on one linkage)
        initial = ahc_cluster(temb, threshold) if len(train) >= 2 else np.zeros(len(train), np.int32)
    if ahc_fails and len(train) >= 2:     # AHCClustering.swift:52-55: a non-zero wrapper status degrades to one cluster per row
        initial = np.arange(len(train), dtype=np.int32)
    if vbx_fails:
        gamma, pi, hard, elbos = vbx_refine_degraded(initial)
    else:
        gamma, pi, hard, elbos = vbx_refine(trho, initial, phi, max_iter, tol, Fa, Fb)
    out = dict(initial=np.asarray(initial), gamma=gamma, pi=pi, hard=hard, elbos=elbos, was_adjusted=False)
    if num_speakers is not None or min_speakers is not None or max_speakers is not None:
        _, lo, hi = speaker_constraints(len(train), num_speakers, min_speakers, max_speakers)
        detected = len(set(np.argmax(gamma, axis=1).tolist())) if gamma.size else int((np.asarray(pi) > 1e-7).sum())
        if detected < lo or detected > hi:
            target = min(max(detected, lo), hi)
            km, cent, _, _ = kmeans_ninit(temb, target, 100, 10, 0)
            out.update(was_adjusted=True, detected=detected, kmeans_clusters=km)
    if not out["was_adjusted"]:
        cent, _ = weighted_centroids(temb, gamma, pi)
    if cent.shape[0] > 1 and constrained and not out["was_adjusted"]:
        assign = constrained_assign(centroid_scores(emb, cent), chunk_indices)
    else:
        assign = assign_cosine(emb, cent)
    out.update(assignments=assign, centroids=cent)
    return out


class SeededRNG:
    """KMeansClustering.SeededRNG (:212-223) + the Swift-stdlib draws built on it (restated, see fa_oracle.c)."""

    def __init__(self, seed: int):
        self._s = C.c_uint64(seed & (2 ** 64 - 1))

    def next(self) -> int:
        return int(lib().fa_oracle_seeded_rng_next(C.byref(self._s)))

    def next_upper_bound(self, bound: int) -> int:
        return int(lib().fa_oracle_rng_upper_bound(C.byref(self._s), bound))

    def shuffled_indices(self, n: int) -> np.ndarray:
        idx = np.zeros(max(n, 1), np.int64)
        lib().fa_oracle_shuffle_indices(C.byref(self._s), n, idx)
        return idx[:n]

    def random_double(self, lo: float, hi: float) -> float:
        """Double.random(in: lo...hi, using:) of the Swift stdlib (FloatingPointRandom.swift, closed range): 53 random bits + 1
        extra value for the closed upper end: rand = next(upperBound: 2^53 + 1); unit = rand == 2^53 ? 1 : rand * 2^-53."""
        r = self.next_upper_bound((1 << 53) + 1)
        unit = 1.0 if r == (1 << 53) else r * (2.0 ** -53)
        return lo + (hi - lo) * unit


def kmeans(emb, num_clusters: int, max_iter: int = 300, seed: int = 0):
    """KMeansClustering.clusterWithCentroids -> (labels int32[n], centroids [k, d], iterations)."""
    x = np.ascontiguousarray(emb, np.float64)
    n, d = (x.shape[0], x.shape[1]) if x.ndim == 2 else (len(x), 0)
    lab = np.zeros(max(n, 1), np.int32)
    cen = np.zeros((max(min(num_clusters, n), 1), max(d, 1)), np.float64)
    k, it = C.c_long(), C.c_long()
    st = lib().fa_oracle_kmeans(x if x.size else np.zeros((1, 1)), n, d, num_clusters, max_iter, seed & (2 ** 64 - 1), lab, cen,
                                C.byref(k), C.byref(it))
    assert st == 0
    return lab[:n], cen[:k.value, :d].copy(), it.value


def kmeans_ninit(emb, num_clusters: int, max_iter: int = 300, n_init: int = 10, base_seed: int = 0):
    """KMeansClustering.clusterWithCentroidsNInit -> (labels, centroids, best run, inertias)."""
    x = np.ascontiguousarray(emb, np.float64)
    n, d = (x.shape[0], x.shape[1]) if x.ndim == 2 else (len(x), 0)
    lab = np.zeros(max(n, 1), np.int32)
    cen = np.zeros((max(min(num_clusters, n), 1), max(d, 1)), np.float64)
    k, best = C.c_long(), C.c_long()
    inert = np.full(max(n_init, 1), np.nan)
    st = lib().fa_oracle_kmeans_ninit(x if x.size else np.zeros((1, 1)), n, d, num_clusters, max_iter, n_init,
                                      base_seed & (2 ** 64 - 1), lab, cen, C.byref(k), C.byref(best), inert)
    assert st == 0
    return lab[:n], cen[:k.value, :d].copy(), best.value, inert


def speaker_constraints(num_embeddings: int, num_speakers=None, min_speakers=None, max_speakers=None):
    """SpeakerCountConstraints.resolve -> (numSpeakers or None, minSpeakers, maxSpeakers)."""
    out = (C.c_long * 3)()
    lib().fa_oracle_speaker_constraints(num_embeddings, num_speakers is not None, num_speakers or 0, min_speakers is not None,
                                        min_speakers or 0, max_speakers is not None, max_speakers or 0, C.byref(out))
    return (None if out[0] < 0 else int(out[0])), int(out[1]), int(out[2])


def tdt_initial_time_index(time_jump, context_frame_adjustment: int) -> int:
    return lib().fa_oracle_tdt_initial_time_index(0 if time_jump is None else 1, 0 if time_jump is None else int(time_jump),
                                                  int(context_frame_adjustment))


def tdt_clamp_probability(v: float) -> float:
    return float(lib().fa_oracle_tdt_clamp_probability(float(v)))


def tdt_greedy(tok, dur_bin, prob, enc_len, audio_frames=None, t0=0, is_last=False, global_offset=0, emit_after=None,
               blank_id=8192, max_symbols=10, max_tokens=150, blank_limit=5, bins=(0, 1, 2, 3, 4), max_out=512):
    """One chunk of TdtDecoderV3.decodeWithTimings over [U, T] joint-decision tables -> dict."""
    tok = np.ascontiguousarray(tok, np.int32)
    dur_bin = np.ascontiguousarray(dur_bin, np.int32)
    prob = np.ascontiguousarray(prob, np.float32)
    U, T = tok.shape
    ot, oti, od = (np.zeros(max(max_out, 1), np.int32) for _ in range(3))
    oc = np.zeros(max(max_out, 1), np.float32)
    cnt, ft, fu = C.c_int(), C.c_int(), C.c_int()
    b = np.ascontiguousarray(bins, np.int32)
    st = lib().fa_oracle_tdt_greedy(tok, dur_bin, prob, U, T, int(enc_len), int(enc_len if audio_frames is None else audio_frames),
                                    int(t0), int(bool(is_last)), int(global_offset), -1 if emit_after is None else int(emit_after),
                                    blank_id, max_symbols, max_tokens, blank_limit, b, b.size, max_out,
                                    ot, oti, od, oc, C.byref(cnt), C.byref(ft), C.byref(fu))
    n = min(cnt.value, max_out)
    return dict(status=st, tokens=ot[:n].copy(), timestamps=oti[:n].copy(), durations=od[:n].copy(), confidences=oc[:n].copy(),
                count=cnt.value, final_time=None if ft.value == -2 ** 31 else ft.value, final_u=fu.value,
                joint_calls=int(lib().fa_oracle_tdt_last_joint_calls()))


def ctc_log_probs(logits, temperature: float = 1.0, blank_bias: float = 0.0, blank_id: int = -1) -> np.ndarray:
    """CtcKeywordSpotter.makeLogProbs (:350-405) on a [T, V] float32 matrix."""
    x = np.ascontiguousarray(logits, np.float32)
    out = np.zeros_like(x)
    for t in range(x.shape[0]):
        lib().fa_oracle_log_softmax_row(x[t], x.shape[1], temperature, blank_bias, blank_id, out[t])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# wire formats (pure Python / numpy restatements; small inputs only)
def wav_pcm16(samples, sample_rate: float, normalize: bool = True) -> bytes:
    """AudioWAV.data (Sources/FluidAudio/Shared/AudioConverter.swift:474-532) in float32 arithmetic."""
    import struct
    x = np.asarray(samples, np.float32)
    mx = np.float32(np.abs(x).max()) if x.size else np.float32(1.0)                  # :486
    norm = (x / mx).astype(np.float32) if (normalize and mx > 0) else x             # :487
    clipped = np.maximum(np.float32(-1.0), np.minimum(np.float32(1.0), norm))       # :493
    pcm = np.trunc((clipped * np.float32(32767)).astype(np.float32)).astype("<i2")  # Int16(Float): toward zero (:494)
    hdr = b"RIFF" + struct.pack("<I", 36 + 2 * pcm.size) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, int(sample_rate),
                                                                                             int(sample_rate * 2), 2, 16)
    return hdr + b"data" + struct.pack("<I", 2 * pcm.size) + pcm.tobytes()


# Character classes of the reference's text readers (Foundation / Swift standard library):
#   CharacterSet.whitespaces = Unicode category Zs + tab; CharacterSet.newlines = U+000A ... U+000D, U+0085, U+2028, U+2029;
#   Character.isWhitespace (the Unicode White_Space property) = CharacterSet.whitespacesAndNewlines = their union.
_WS = "\t \u00a0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u202f\u205f\u3000"
_NL = "\n\x0b\x0c\r\u0085\u2028\u2029"
_RE_NL = re.compile("[" + _NL + "]")
_RE_WSNL = re.compile("[" + _WS + _NL + "]+")


def rttm_parse(text: str, strict: bool = True):
    """RTTMParser.loadSegments (RTTMParser.swift:22-63) / SortformerBenchmark.loadRTTMGroundTruth (:681-731) -> list of
    (speaker, start float32, end float32); raises ValueError(line) in strict mode.  Lines = components(separatedBy: .newlines),
    trimmed with .whitespaces; fields split at white space (:36 / :703-705, empty fields dropped); numbers = Float(String)."""
    out = []
    for raw in _RE_NL.split(text):
        line = raw.strip(_WS)
        if not line or (strict and line.startswith("#")):
            continue
        f = [x for x in _RE_WSNL.split(line) if x]
        ok = len(f) >= 8 and f[0] == "SPEAKER"
        if ok:
            try:
                start, dur = _swift_float(f[3]), _swift_float(f[4])
            except ValueError:
                ok = False
        if not ok:
            if strict:
                raise ValueError(line)
            continue
        out.append((f[7], float(start), float(np.float32(start + dur))))
    if strict:
        out.sort(key=lambda s: s[1])      # Python's sort is stable, like the reference's (:62)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CTC prefix beam search with ARPA language model (pure Python restatement; small cases only)
WORD_BOUNDARY = "▁"          # ASRConstants.sentencePieceWordBoundary


_RE_NAN_PAYLOAD = re.compile(r"[+-]?nan\([0-9a-z]*\)\Z", re.I)


def _swift_float(s: str):
    """Float(String) of the Swift standard library: the whole string is one number — decimal or hexadecimal ("0x1.8p3"), "inf" /
    "infinity" / "nan" in any case, optional sign; no surrounding white space, no digit separators, ASCII only.  (Python's float()
    also takes "1_0", Arabic-Indic digits and surrounding blanks, and no hexadecimal: hence the checks.)"""
    if not s or not s.isascii() or "_" in s or any(ord(c) <= 32 or ord(c) == 127 for c in s):
        raise ValueError(s)
    body = s.lstrip("+-").lower()
    with np.errstate(over="ignore"):
        if body.startswith("0x"):
            try:
                return np.float32(float.fromhex(s))
            except OverflowError:
                return np.float32(-np.inf if s.startswith("-") else np.inf)
        if _RE_NAN_PAYLOAD.match(s):
            return np.float32(np.nan)
        return np.float32(s)


class ARPALanguageModel:
    """ARPALanguageModel (Sources/FluidAudio/ASR/Parakeet/SlidingWindow/CTC/ARPALanguageModel.swift:16-104): unigrams and
    bigrams of a plain-text ARPA file, log10 -> natural log in float32."""
    LOG10_TO_NAT = np.float32(np.log(10.0))
    UNK_LOG_PROB = np.float32(-23.026)

    def __init__(self):
        self.unigrams, self.bigrams = {}, {}

    @classmethod
    def parse(cls, text: str):
        lm, section = cls(), ""
        for raw in text.split("\n"):                                                    # the reader cuts at the byte \\n only (:126)
            line = raw.strip(_WS + _NL)                                                  # .whitespacesAndNewlines (:131)
            if not line or line.startswith("\\data\\"):
                continue
            if line == "\\end\\":
                break
            if line.startswith("\\"):
                section = line
                continue
            if line.startswith("ngram "):
                continue
            parts = line.split("\t")
            try:
                prob = np.float32(_swift_float(parts[0]) * cls.LOG10_TO_NAT)
            except ValueError:
                continue                                                             # malformed line skipped (:64-67)

            def backoff(i):
                try:
                    return np.float32(_swift_float(parts[i]) * cls.LOG10_TO_NAT) if len(parts) > i else np.float32(0.0)
                except ValueError:
                    return np.float32(0.0) * cls.LOG10_TO_NAT
            if section == "\\1-grams:" and len(parts) >= 2:
                lm.unigrams[parts[1]] = (prob, backoff(2))
            elif section == "\\2-grams:" and len(parts) >= 3:
                lm.bigrams.setdefault(parts[1], {})[parts[2]] = (prob, backoff(3))
        return lm

    def score(self, word: str, prev):                                                # :98-103
        if prev is not None and word in self.bigrams.get(prev, {}):
            return self.bigrams[prev][word][0]
        bo = self.unigrams[prev][1] if (prev is not None and prev in self.unigrams) else np.float32(0.0)
        return np.float32(bo + (self.unigrams[word][0] if word in self.unigrams else self.UNK_LOG_PROB))


def log_add_exp(a, b):                                                               # CtcDecoder.swift:279-284
    a, b = np.float32(a), np.float32(b)
    if a == -np.inf:
        return b
    if b == -np.inf:
        return a
    m = max(a, b)
    # evaluated in double and rounded once: within 1 ulp of the reference's Float exp/log (Darwin libm, not reproducible
    # bit for bit anyway) and identical between this restatement and the device
    return np.float32(np.float64(m) + np.log(np.exp(np.float64(np.float32(a - m))) + np.exp(np.float64(np.float32(b - m)))))


def ctc_beam_search(log_probs, vocabulary: dict, lm=None, beam_width=100, lm_weight=0.3, word_bonus=0.0, blank_id=1024,
                    token_candidates=40):
    """ctcBeamSearch (CtcDecoder.swift:118-241) -> (token ids of the best prefix, its total score).  The reference iterates
    Swift dictionaries (arbitrary order); every quantity is order-independent except ties, which are resolved HERE as:
    top tokens by (log-prob desc, index asc); beams in rank order, each followed by its extensions in token order; stable
    sort by total for the pruning; first maximum at the end."""
    lp = np.asarray(log_probs, np.float32)
    if lp.size == 0 or lp.shape[1] == 0:
        return [], None
    V = lp.shape[1]
    lm_weight, word_bonus = np.float32(lm_weight), np.float32(word_bonus)
    NEG = np.float32(-np.inf)
    beams = {(): dict(prefix=(), pb=np.float32(0.0), pnb=NEG, lm=np.float32(0.0), pieces=(), prev=None)}
    total_ac = lambda b: log_add_exp(b["pb"], b["pnb"])
    total = lambda b: np.float32(total_ac(b) + b["lm"])
    for frame in lp:
        blank_lp = frame[blank_id] if 0 <= blank_id < V else NEG
        top = sorted((v for v in range(V) if v != blank_id), key=lambda v: (-frame[v], v))[:token_candidates]
        new = {}

        def merge(b):
            e = new.get(b["prefix"])
            if e is not None:
                e["pb"], e["pnb"] = log_add_exp(e["pb"], b["pb"]), log_add_exp(e["pnb"], b["pnb"])
            else:
                new[b["prefix"]] = b
        for beam in list(beams.values()):
            prev_total = total_ac(beam)
            merge(dict(beam, pb=np.float32(prev_total + blank_lp), pnb=NEG))
            for v in top:
                tlp = frame[v]
                last = beam["prefix"][-1] if beam["prefix"] else None
                piece = vocabulary.get(v, "")
                pieces, prev, delta = beam["pieces"], beam["prev"], np.float32(0.0)
                if lm is not None and piece.startswith(WORD_BOUNDARY):
                    done = "".join(pieces)
                    if done:
                        delta = np.float32(np.float32(lm_weight * lm.score(done, prev)) + word_bonus)
                        prev = done
                    stripped = piece[1:]
                    pieces = (stripped,) if stripped else ()
                elif lm is not None:
                    pieces = pieces + (piece,)
                ext = dict(prefix=beam["prefix"] + (v,), pb=NEG, lm=np.float32(beam["lm"] + delta), pieces=pieces, prev=prev)
                if last == v:
                    merge(dict(beam, pb=NEG, pnb=np.float32(beam["pnb"] + tlp)))
                    merge(dict(ext, pnb=np.float32(beam["pb"] + tlp)))
                else:
                    merge(dict(ext, pnb=np.float32(prev_total + tlp)))
        # deterministic candidate order: existing prefixes keep their beam's position, new ones follow their parent
        order = sorted(new.values(), key=lambda b: -total(b))                        # Python's sort is stable
        beams = {b["prefix"]: b for b in order[:beam_width]}
    best, best_total = None, None
    for b in beams.values():
        t = total(b)
        if lm is not None:
            word = "".join(b["pieces"])
            if word:
                t = np.float32(total_ac(b) + np.float32(b["lm"] + np.float32(np.float32(lm_weight * lm.score(word, b["prev"])) + word_bonus)))
        if best is None or t > best_total:
            best, best_total = b, t
    return list(best["prefix"]), float(best_total)


def decode_ctc_token_ids(ids, vocabulary: dict) -> str:                             # CtcDecoder.swift:289-294
    return "".join(vocabulary[i] for i in ids if i in vocabulary).replace(WORD_BOUNDARY, " ").strip(" \t")


# ------------------------------------------------------------------ float64 evaluation of the mel path
def mel_f64(audio, cfg: MelConfig = MelConfig(), last: float = 0.0, padding: str = "center", expected_frames: int | None = None):
    """The reference's formula (AudioMelSpectrogram.swift:185-292 / :325-456 / :132-178) evaluated in float64 on the
    reference's own fp32 constants: the Hann window and Slaney filterbank are the fp32 tables of :553-642 (part of the
    definition), the pre-emphasis coefficient and log floor are the fp32 numbers, the INPUT is fp32 — but every sum and
    product, the DFT and the log run in float64.  It is what an fp32 implementation is an approximation OF, so
    |fp32 path - mel_f64| measures that path's arithmetic error (used as the 1e-4 gate of north_star for both the fp32
    restatement and the device).  Returns [T, n_mels] float64 (T = reference frame count; no padTo padding).
    padding: 'center' (computeFlat/-Transposed .center), 'prepadded' (:345), 'legacy' (compute(), :132-178)."""
    a = np.ascontiguousarray(audio, np.float32).astype(np.float64)
    n_fft, hop, win = cfg.n_fft, cfg.hop, cfg.win
    w = hann(win, cfg.window_periodic).astype(np.float64)
    fb = slaney_filterbank(n_fft, cfg.n_mels, cfg.sample_rate).astype(np.float64)
    if padding == "legacy":
        T = 1 + int((a.size - win) / hop) if a.size >= win else 0
        y, pad, off = a, 0, 0
    else:
        T = mel_frames(cfg, a.size, prepadded=(padding == "prepadded"))
        p = float(np.float32(cfg.preemph))
        if p != 0.0:                                                   # :211,:219-225 (:363-371: plain copy when preemph == 0)
            y = a.copy()
            if a.size:
                y[0] = a[0] - p * float(np.float32(last))
                y[1:] = a[1:] - p * a[:-1]
        else:
            y = a
        pad = n_fft // 2 if padding == "center" else 0
        off = (n_fft - win) // 2                                       # :234
    if expected_frames is not None and a.size > 0:
        T = max(int(expected_frames), 0)
    if T <= 0:
        return np.zeros((0, cfg.n_mels))
    buf = np.zeros(pad + max(y.size, 0) + pad + n_fft + T * hop)      # zero beyond the signal (= truncated windows, :412)
    buf[pad:pad + y.size] = y
    idx = np.arange(T)[:, None] * hop + off + np.arange(win)[None, :]
    frames = np.zeros((T, n_fft))
    frames[:, off:off + win] = buf[idx] * w[None, :]
    power = np.abs(np.fft.rfft(frames, axis=1)) ** 2
    v = power @ fb.T
    floor = float(np.float32(cfg.log_floor))
    return np.log(np.maximum(v, floor)) if cfg.floor_clamped else np.log(v + floor)


def mel_f64_error(got, ref64) -> float:
    """max over elements of |got - f64| / max(|f64|, 1e-2): pure relative error wherever |log-mel| >= 1e-2."""
    got, ref64 = np.asarray(got, np.float64), np.asarray(ref64, np.float64)
    return float(np.max(np.abs(got - ref64) / np.maximum(np.abs(ref64), 1e-2))) if got.size else 0.0


def unified_mel_features_f64(window, valid_count: int, n_mels: int = 128, hop: int = 160):
    """UnifiedMelExtractor.features (:52-113) in float64 on the fp32 input: mel_f64 + per-feature mean / unbiased std over
    the valid frames (+1e-5), frames >= valid -> 0.  Returns ([n_mels, totalFrames] float64, valid frames)."""
    w = np.ascontiguousarray(window, np.float32)
    total = w.size // hop + 1                                             # :29
    m = mel_f64(w, MelConfig(n_mels=n_mels), expected_frames=total).T     # [n_mels, T]
    valid = min(valid_count // hop, total)                                # :66
    out = np.zeros_like(m)
    if valid > 0:
        x = m[:, :valid]
        mean = x.sum(axis=1, keepdims=True) / valid
        var = ((x - mean) ** 2).sum(axis=1, keepdims=True) / max(valid - 1, 1)
        out[:, :valid] = (x - mean) / (np.sqrt(var) + float(np.float32(1e-5)))
    return out, valid
