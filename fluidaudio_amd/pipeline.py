"""The arithmetic of ``OfflineDiarizerManager.cluster`` (reference:
Sources/FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:270-375) on precomputed embeddings — the "embeddings in
-> final per-embedding cluster ids out" path of BASELINE config 5.  ``cluster_embeddings`` is ONE call into the library
(``fa_offline_cluster``, csrc/offline.hip: inputs go up once, the intermediates stay in HBM); ``cluster_embeddings_stagewise``
is the same composition made of the single-stage entries (every intermediate crosses PCIe), kept because it exposes the
intermediate results (AHC labels, VBx posteriors) the tests compare with the CPU restatement:

    select finite embeddings (:591-611) -> AHC (AHCClustering.cluster, threshold 0.6) -> VBx refine (Fa 0.07, Fb 0.8,
    <= 20 iterations) -> gamma-weighted centroids of the active speakers (:613-691) -> cosine scores (:789-798) ->
    constrained per-chunk assignment (ConstrainedClusterAssignment, default) or plain argmax (:800-822).

Speaker-count constraints (numSpeakers / minSpeakers / maxSpeakers, :308-326) route through
``VBxClustering.refine_with_constraints`` and the batched K-Means fallback; as in the reference (:355-358) the constrained
assignment is skipped when the count was forced.  Segmentation, embedding extraction and PLDA are CoreML models in the
reference (out of scope)."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _lib as L
from .ahc import AHCClustering
from .kmeans import SpeakerCountConstraints
from .post import ConstrainedClusterAssignment, assign_embeddings, centroid_scores, compute_centroids
from .vbx import VBxClustering, VBxOutput


@dataclass
class OfflineClusteringConfig:  # OfflineDiarizerTypes.swift:155-163,189-192
    clustering_threshold: float = 0.6
    warm_start_fa: float = 0.07
    warm_start_fb: float = 0.8
    max_vbx_iterations: int = 20
    convergence_tolerance: float = 1e-4
    constrained_assignment: bool = True
    num_speakers: int | None = None      # OfflineDiarizerTypes.swift clustering.numSpeakers / minSpeakers / maxSpeakers
    min_speakers: int | None = None
    max_speakers: int | None = None

    def validate(self) -> None:
        """The clustering / VBx guards of OfflineDiarizerConfig.validate (OfflineDiarizerTypes.swift:357-408; the reference throws
        OfflineDiarizationError.invalidConfiguration with these messages before any audio is touched)."""
        t = self.clustering_threshold
        if not (t > 0 and t <= 2.0):                         # distances between unit rows live in [0, 2] (:358-364); NaN fails too
            raise ValueError(f"invalidConfiguration: clustering.threshold must be within (0, 2], got {t}")
        if not (self.warm_start_fa > 0 and self.warm_start_fb > 0):
            raise ValueError(f"invalidConfiguration: clustering warm-start Fa/Fb must be positive (Fa={self.warm_start_fa}, Fb={self.warm_start_fb})")
        if not self.max_vbx_iterations > 0:
            raise ValueError(f"invalidConfiguration: maxVBxIterations must be > 0, got {self.max_vbx_iterations}")
        if not self.convergence_tolerance > 0:
            raise ValueError("invalidConfiguration: convergenceTolerance must be positive")


@dataclass
class ClusteringResult:
    assignments: list
    centroids: np.ndarray
    initial_clusters: list
    vbx: VBxOutput
    training_indices: list = field(default_factory=list)
    timings: dict = field(default_factory=dict)
    info: dict = field(default_factory=dict)


def select_training_embeddings(embedding256) -> list:
    """selectTrainingEmbeddings (:591-611): indices of the rows without NaN/Inf; all rows if none qualifies."""
    e = np.asarray(embedding256)
    ok = np.isfinite(e).all(axis=1) if e.ndim == 2 and e.size else np.zeros(len(e), bool)
    sel = np.nonzero(ok)[0].tolist()
    return sel if sel else list(range(len(e)))


def cluster_embeddings(embedding256, rho128, chunk_indices, phi, config: OfflineClusteringConfig | None = None,
                       ctx: L.Context | None = None, intermediates: bool = False) -> ClusteringResult:
    """One device-resident call (fa_offline_cluster).  ``initial_clusters`` / ``vbx`` of the result are not populated (they never
    leave the device) unless ``intermediates`` asks for copies (fa_offline_cluster_ex: AHC labels in ``initial_clusters``, VBx hard
    labels and ELBOs in ``info["vbx_hard"]`` / ``info["elbos"]``); ``timings`` carries the library's per-stage wall-clock and
    ``info`` its counters."""
    import ctypes as C
    cfg = config or OfflineClusteringConfig()
    cfg.validate()
    ctx = ctx or L.default_context()
    on_device = hasattr(embedding256, "data_ptr")          # torch CUDA tensors: device_pointers = 1, nothing is uploaded
    if on_device:
        emb, rho = embedding256, rho128
        assert emb.is_cuda and emb.is_contiguous() and emb.dtype.itemsize == 4 and emb.dim() == 2
        if emb.shape[0] == 0:
            raise ValueError("noSpeechDetected")
        n, d = int(emb.shape[0]), int(emb.shape[1])
        rd = int(rho.shape[1]) if rho is not None and rho.dim() == 2 and rho.numel() else 0
        if rd:
            assert rho.is_cuda and rho.is_contiguous() and rho.dtype.itemsize == 8 and int(rho.shape[0]) == n
        emb_ptr, rho_ptr = emb.data_ptr(), (rho.data_ptr() if rd else None)
    else:
        emb = np.ascontiguousarray(embedding256, np.float32)
        if emb.ndim != 2 or emb.shape[0] == 0:
            raise ValueError("noSpeechDetected")                                 # :281-283
        n, d = emb.shape
        rho = np.ascontiguousarray(rho128, np.float64)
        rd = rho.shape[1] if rho.ndim == 2 and rho.size else 0
        emb_ptr, rho_ptr = emb.ctypes.data, (rho.ctypes.data if rd else None)
    ph = np.ascontiguousarray(phi, np.float64)
    if rd and ph.size != rd:
        ph = np.ones(rd)                                                      # dimension mismatch -> identity (VBxClustering.swift:72-76)
    chunks = np.ascontiguousarray(chunk_indices, np.int32)
    c = L.OfflineClusterConfig()
    L.lib().fa_offline_cluster_default_config(C.byref(c))
    c.clustering_threshold, c.warm_start_fa, c.warm_start_fb = cfg.clustering_threshold, cfg.warm_start_fa, cfg.warm_start_fb
    c.max_vbx_iterations, c.convergence_tolerance = cfg.max_vbx_iterations, cfg.convergence_tolerance
    c.constrained_assignment = int(cfg.constrained_assignment)
    c.num_speakers = -1 if cfg.num_speakers is None else cfg.num_speakers
    c.min_speakers = -1 if cfg.min_speakers is None else cfg.min_speakers
    c.max_speakers = -1 if cfg.max_speakers is None else cfg.max_speakers
    labels = np.zeros(n, np.int32)
    cap = 256
    cen = np.zeros((cap, d), np.float64)
    k, info = C.c_int32(), L.OfflineClusterInfo()
    if intermediates:
        ahc_lab, hard, elbos = np.full(n, -1, np.int32), np.full(n, -1, np.int32), np.zeros(max(cfg.max_vbx_iterations, 1))
        ctx.check(L.lib().fa_offline_cluster_ex(ctx.handle, emb_ptr, n, d, rho_ptr, rd, chunks.ctypes.data,
                                                ph.ctypes.data if rd else None, C.byref(c), int(on_device), labels.ctypes.data, cen.ctypes.data, cap, C.byref(k),
                                                C.byref(info), ahc_lab.ctypes.data, hard.ctypes.data, elbos.ctypes.data), "fa_offline_cluster_ex")
    else:
        ctx.check(L.lib().fa_offline_cluster(ctx.handle, emb_ptr, n, d, rho_ptr, rd, chunks.ctypes.data,
                                             ph.ctypes.data if rd else None, C.byref(c), int(on_device), labels.ctypes.data, cen.ctypes.data, cap, C.byref(k),
                                             C.byref(info)), "fa_offline_cluster")
    t = {"inputs_s": info.inputs_s, "ahc_s": info.ahc_s, "vbx_s": info.vbx_s, "assign_s": info.assign_s, "total_s": info.total_s}
    res = ClusteringResult(labels if intermediates else np.asarray(labels).tolist(), cen[:k.value].copy(), [], None, [], t)
    res.info = {f: getattr(info, f) for f, _ in info._fields_ if f != "ahc"}
    res.info["ahc"] = info.ahc.as_dict()
    if intermediates:
        nt = int(info.training_rows)
        res.initial_clusters = ahc_lab[:nt]
        res.info["vbx_hard"] = hard[:nt]
        res.info["elbos"] = elbos[:int(info.vbx_iterations)].copy()
    return res


class _DevArray:
    """A torch CUDA tensor behind the few ndarray attributes cluster_embeddings_batch reads (shape / ndim / size / ctypes.data = the device address)."""

    class _Ptr:
        def __init__(self, p):
            self.data = p

    def __init__(self, t):
        self.t = t
        self.shape = tuple(t.shape) if t is not None else (0,)
        self.ndim = len(self.shape)
        self.size = int(t.numel()) if t is not None else 0
        self.ctypes = _DevArray._Ptr(int(t.data_ptr()) if t is not None and t.numel() else 0)


def _c_config(cfg: OfflineClusteringConfig):
    import ctypes as C
    c = L.OfflineClusterConfig()
    L.lib().fa_offline_cluster_default_config(C.byref(c))
    c.clustering_threshold, c.warm_start_fa, c.warm_start_fb = cfg.clustering_threshold, cfg.warm_start_fa, cfg.warm_start_fb
    c.max_vbx_iterations, c.convergence_tolerance = cfg.max_vbx_iterations, cfg.convergence_tolerance
    c.constrained_assignment = int(cfg.constrained_assignment)
    c.num_speakers = -1 if cfg.num_speakers is None else cfg.num_speakers
    c.min_speakers = -1 if cfg.min_speakers is None else cfg.min_speakers
    c.max_speakers = -1 if cfg.max_speakers is None else cfg.max_speakers
    return c


def cluster_embeddings_batch(recordings, phi, config: OfflineClusteringConfig | None = None, ctx: L.Context | None = None):
    """Several recordings through the clustering stage in one call (fa_offline_cluster_batch: their merge chains advance together).
    recordings: iterable of (embedding256 [n, d] float32, rho128 [n, rho_dim] float64, chunk_indices [n]) with common d / rho_dim.
    Embeddings / rho given as torch CUDA tensors (all recordings, or none) go through fa_offline_cluster_batch_dev: nothing is uploaded.
    Returns (statuses, [ClusteringResult | None]) — per recording identical to cluster_embeddings()."""
    import ctypes as C
    cfg = config or OfflineClusteringConfig()
    cfg.validate()
    ctx = ctx or L.default_context()
    recordings = list(recordings)
    on_device = bool(recordings) and all(hasattr(e, "data_ptr") for e, _, _ in recordings)
    if on_device:
        for e, r, _ in recordings:
            assert e.is_cuda and e.is_contiguous() and e.dtype.itemsize == 4 and e.dim() == 2
            assert r is None or (r.is_cuda and r.is_contiguous() and r.dtype.itemsize == 8)
        recs = [(_DevArray(e), _DevArray(r), np.ascontiguousarray(c, np.int32)) for e, r, c in recordings]
    else:
        recs = [(np.ascontiguousarray(e, np.float32), np.ascontiguousarray(r, np.float64), np.ascontiguousarray(c, np.int32)) for e, r, c in recordings]
    k = len(recs)
    if k == 0:
        return [], []
    d = recs[0][0].shape[1]
    rd = recs[0][1].shape[1] if recs[0][1].ndim == 2 and recs[0][1].size else 0
    ph = np.ascontiguousarray(phi, np.float64)
    if rd and ph.size != rd:
        ph = np.ones(rd)
    cap = 256
    labels = [np.zeros(max(e.shape[0], 1), np.int32) for e, _, _ in recs]
    cens = [np.zeros((cap, d), np.float64) for _ in recs]
    dummy = np.zeros(4, np.float32)
    P = C.c_void_p * k
    ep = P(*[e.ctypes.data if e.size else dummy.ctypes.data for e, _, _ in recs])
    rp = P(*[r.ctypes.data if r.size else dummy.ctypes.data for _, r, _ in recs])
    cp = P(*[c.ctypes.data if c.size else dummy.ctypes.data for _, _, c in recs])
    lp = P(*[x.ctypes.data for x in labels])
    zp = P(*[x.ctypes.data for x in cens])
    ns = (C.c_int64 * k)(*[e.shape[0] for e, _, _ in recs])
    kc = (C.c_int32 * k)()
    infos = (L.OfflineClusterInfo * k)()
    st = (C.c_int32 * k)()
    c = _c_config(cfg)
    entry = L.lib().fa_offline_cluster_batch_dev if on_device else L.lib().fa_offline_cluster_batch
    entry(ctx.handle, k, ep, ns, d, rp if rd else None, rd, cp, ph.ctypes.data if rd else None, C.byref(c), lp, zp, cap, kc, infos, st)
    out = []
    for i in range(k):
        if st[i] != L.SUCCESS:
            out.append(None)
            continue
        info = infos[i]
        t = {"inputs_s": info.inputs_s, "ahc_s": info.ahc_s, "vbx_s": info.vbx_s, "assign_s": info.assign_s, "total_s": info.total_s}
        res = ClusteringResult(labels[i][:recs[i][0].shape[0]].tolist(), cens[i][:kc[i]].copy(), [], None, [], t)
        res.info = {f: getattr(info, f) for f, _ in info._fields_ if f != "ahc"}
        res.info["ahc"] = info.ahc.as_dict()
        out.append(res)
    return [int(v) for v in st], out


def cluster_embeddings_stagewise(embedding256, rho128, chunk_indices, phi, config: OfflineClusteringConfig | None = None,
                                 ctx: L.Context | None = None) -> ClusteringResult:
    import time
    cfg = config or OfflineClusteringConfig()
    cfg.validate()
    ctx = ctx or L.default_context()
    t = {}
    emb = np.asarray(embedding256, np.float32).astype(np.float64)          # Float -> Double widening (:286)
    rho = np.ascontiguousarray(rho128, np.float64)
    if emb.shape[0] == 0:
        raise ValueError("noSpeechDetected")                                 # :281-283
    train = select_training_embeddings(embedding256)
    temb, trho = emb[train], rho[train]
    t0 = time.perf_counter()
    if len(train) >= 2:
        initial = AHCClustering(ctx=ctx).cluster(temb, cfg.clustering_threshold)   # :301-306
    else:
        initial = [0] * len(train)
    t["ahc_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if trho.size and initial:
        has = cfg.num_speakers is not None or cfg.min_speakers is not None or cfg.max_speakers is not None   # :309-312
        cons = SpeakerCountConstraints.resolve(len(train), cfg.num_speakers, cfg.min_speakers, cfg.max_speakers) if has else None
        vbx = VBxClustering(phi, cfg.max_vbx_iterations, cfg.convergence_tolerance, cfg.warm_start_fa, cfg.warm_start_fb,
                            ctx=ctx).refine_with_constraints(trho, temb, initial, cons)    # :328-333
    else:
        vbx = VBxOutput(np.zeros((0, 0)), np.zeros(0), [list(initial)], [], (max(initial) + 1) if initial else 0, [])
    t["vbx_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    centroids = np.zeros((0, emb.shape[1]))
    if vbx.was_adjusted and len(vbx.centroids):                              # K-Means centroids as they are (:622-629)
        centroids = np.asarray(vbx.centroids, np.float64)
    elif vbx.gamma.size and vbx.pi.size:
        centroids, _ = compute_centroids(temb, vbx.gamma, vbx.pi, ctx=ctx)  # :613-684
    if centroids.shape[0] == 0:                                              # computeCentroidsFromClusters (:686-) fallback
        labs = np.asarray(initial)
        # sequential sums in row order (cblas_daxpy per row, :700-728), then one division: numpy's mean() sums pairwise
        centroids = np.stack([np.cumsum(temb[labs == k], axis=0)[-1] / float((labs == k).sum()) for k in sorted(set(initial))]) if len(initial) else centroids
    use_constrained = cfg.constrained_assignment and not vbx.was_adjusted and centroids.shape[0] > 1  # :355-358
    if use_constrained:
        scores = centroid_scores(emb, centroids, ctx=ctx)
        assignments = ConstrainedClusterAssignment.assign(scores, chunk_indices, ctx=ctx)
    else:
        assignments = assign_embeddings(emb, centroids, ctx=ctx)
    t["assign_s"] = time.perf_counter() - t0
    return ClusteringResult(assignments, centroids, list(initial), vbx, train, t)
