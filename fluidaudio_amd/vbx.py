"""Host-side mirror of ``VBxClustering.refine`` (reference:
Sources/FluidAudio/Diarizer/Offline/Clustering/VBxClustering.swift:41-165) over the HIP C ABI.
Config defaults follow OfflineDiarizerTypes.swift:155-163,189-192 (Fa 0.07, Fb 0.8, 20 iterations, tol 1e-4).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L


@dataclass
class VBxOutput:
    gamma: np.ndarray
    pi: np.ndarray
    hard_clusters: list
    centroids: list = field(default_factory=list)
    num_clusters: int = 0
    elbos: list = field(default_factory=list)
    was_adjusted: bool = False                 # OfflineDiarizerTypes.swift VBxOutput.wasAdjusted
    original_cluster_count: int | None = None

    ACTIVE_CLUSTER_EPSILON = 1e-7

    @property
    def active_cluster_count(self) -> int:     # OfflineDiarizerTypes.swift:675-678
        pi = np.asarray(self.pi)
        return self.num_clusters if pi.size == 0 else int((pi > self.ACTIVE_CLUSTER_EPSILON).sum())

    @property
    def assigned_cluster_count(self) -> int:   # OfflineDiarizerTypes.swift:687-702: clusters that win some row's first-max argmax
        g = np.asarray(self.gamma)
        if g.size == 0:
            return self.active_cluster_count
        return len(set(np.argmax(g, axis=1).tolist()))


class VBxClustering:
    def __init__(self, phi_parameters, max_iterations: int = 20, convergence_tolerance: float = 1e-4,
                 warm_start_fa: float = 0.07, warm_start_fb: float = 0.8, ctx: L.Context | None = None):
        self.phi = np.ascontiguousarray(phi_parameters, np.float64)
        self.max_iterations, self.tol = max_iterations, convergence_tolerance
        self.fa, self.fb = warm_start_fa, warm_start_fb
        self._ctx = ctx

    def refine(self, rho_features, initial_clusters) -> VBxOutput:
        rho = np.ascontiguousarray(rho_features, np.float64)
        if rho.size == 0 or rho.ndim != 2 or rho.shape[1] == 0:
            return VBxOutput(np.zeros((0, 0)), np.zeros(0), [], [], 0, [])  # :45-67
        T, D = rho.shape
        init = np.ascontiguousarray(initial_clusters, np.int32)
        phi = self.phi if self.phi.size == D else np.ones(D)  # dimension mismatch -> identity (:72-76)
        ctx = self._ctx or L.default_context()
        S = max(1, L.lib().fa_vbx_speaker_count(init.ctypes.data, T))
        gamma = np.zeros((T, S), np.float64)
        pi = np.zeros(S, np.float64)
        hard = np.zeros(T, np.int32)
        elbos = np.zeros(max(self.max_iterations, 1), np.float64)
        it, ns = C.c_int32(), C.c_int32()
        ctx.check(L.lib().fa_vbx_refine(ctx.handle, rho.ctypes.data, T, D, init.ctypes.data, phi.ctypes.data, self.fa,
                                        self.fb, self.max_iterations, self.tol, gamma.ctypes.data, pi.ctypes.data,
                                        hard.ctypes.data, elbos.ctypes.data, C.byref(it), C.byref(ns)), "fa_vbx_refine")
        return VBxOutput(gamma, pi, [np.asarray(hard).tolist()], [], S, [float(v) for v in elbos[:it.value]])

    def refine_with_constraints(self, rho_features, training_embeddings, initial_clusters, constraints) -> VBxOutput:
        """refineWithConstraints (:685-733): when the clusters the posteriors actually use fall outside [min, max] speakers,
        re-cluster the training embeddings with best-of-10 K-Means (<= 100 iterations, seeds 0..9) to the nearest bound."""
        from .kmeans import KMeansClustering
        out = self.refine(rho_features, initial_clusters)
        if constraints is None:
            return out
        detected = out.assigned_cluster_count
        if not constraints.needs_adjustment(detected):
            return out
        target = constraints.target_count(detected)
        clusters, centroids = KMeansClustering.cluster_with_centroids_n_init(training_embeddings, target, 100, 10, 0,
                                                                             ctx=self._ctx)
        return VBxOutput(out.gamma, out.pi, [clusters], centroids, target, out.elbos, True, detected)
