"""Host-side mirror of ``KMeansClustering`` and ``SpeakerCountConstraints`` (reference:
Sources/FluidAudio/Diarizer/Offline/Clustering/KMeansClustering.swift:39-224, SpeakerCountConstraints.swift:6-77) over the HIP
C ABI (csrc/kmeans.hip).  The n_init runs advance together on the device; the random draws are the reference's LCG through
the Swift standard library's bounded draw (restated in the library, see include/fluidaudio_hip.h)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


class SeededRNG:
    """KMeansClustering.SeededRNG (:212-223)."""

    def __init__(self, seed: int):
        self._state = C.c_uint64(seed & (2 ** 64 - 1))

    def next(self) -> int:
        return int(L.lib().fa_seeded_rng_next(C.byref(self._state)))

    def next_below(self, upper_bound: int) -> int:
        """RandomNumberGenerator.next(upperBound:) of the Swift standard library."""
        return int(L.lib().fa_seeded_rng_below(C.byref(self._state), upper_bound))


def _matrix(embeddings):
    x = np.ascontiguousarray(embeddings, np.float64)
    if x.ndim == 1:
        return x.reshape(len(x), 0), len(x), 0
    return x, x.shape[0], x.shape[1]


class KMeansClustering:
    @staticmethod
    def cluster_with_centroids(embeddings, num_clusters: int, max_iterations: int = 300, seed: int | None = None,
                               ctx: L.Context | None = None):
        """clusterWithCentroids (:39-91) -> (clusters list[int], centroids ndarray [k, d])."""
        x, n, d = _matrix(embeddings)
        if n == 0:
            return [], np.zeros((0, d))
        ctx = ctx or L.default_context()
        labels = np.zeros(n, np.int32)
        cen = np.zeros((max(min(num_clusters, n), 1), max(d, 1)), np.float64)
        k, it = C.c_int32(), C.c_int32()
        ctx.check(L.lib().fa_kmeans_cluster(ctx.handle, x.ctypes.data, n, d, num_clusters, max_iterations, (seed or 0) & (2 ** 64 - 1),
                                            labels.ctypes.data, cen.ctypes.data, C.byref(k), C.byref(it)), "fa_kmeans_cluster")
        return np.asarray(labels).tolist(), cen[:k.value, :d].copy()

    @staticmethod
    def cluster(embeddings, num_clusters: int, max_iterations: int = 300, seed: int | None = None, ctx: L.Context | None = None) -> list:
        return KMeansClustering.cluster_with_centroids(embeddings, num_clusters, max_iterations, seed, ctx)[0]

    @staticmethod
    def cluster_with_centroids_n_init(embeddings, num_clusters: int, max_iterations: int = 300, n_init: int = 10, base_seed: int = 0,
                                      ctx: L.Context | None = None, details: dict | None = None):
        """clusterWithCentroidsNInit (:99-129) -> (clusters, centroids); ``details`` receives best_run and inertias."""
        x, n, d = _matrix(embeddings)
        if n == 0:
            return [], np.zeros((0, d))
        ctx = ctx or L.default_context()
        labels = np.zeros(n, np.int32)
        cen = np.zeros((max(min(num_clusters, n), 1), max(d, 1)), np.float64)
        k, best = C.c_int32(), C.c_int32()
        inert = np.full(max(n_init, 1), np.nan)
        ctx.check(L.lib().fa_kmeans_cluster_ninit(ctx.handle, x.ctypes.data, n, d, num_clusters, max_iterations, n_init,
                                                  base_seed & (2 ** 64 - 1), labels.ctypes.data, cen.ctypes.data, C.byref(k), C.byref(best),
                                                  inert.ctypes.data), "fa_kmeans_cluster_ninit")
        if details is not None:
            details.update(best_run=best.value, inertias=inert)
        return np.asarray(labels).tolist(), cen[:k.value, :d].copy()


@dataclass(frozen=True)
class SpeakerCountConstraints:
    num_speakers: int | None
    min_speakers: int
    max_speakers: int

    @staticmethod
    def resolve(num_embeddings: int, num_speakers: int | None = None, min_speakers: int | None = None,
                max_speakers: int | None = None) -> "SpeakerCountConstraints":
        def opt(v):
            return None if v is None else C.byref(C.c_int64(int(v)))
        out = (C.c_int64 * 3)()
        L.lib().fa_speaker_constraints_resolve(num_embeddings, opt(num_speakers), opt(min_speakers), opt(max_speakers), C.byref(out))
        return SpeakerCountConstraints(None if out[0] < 0 else int(out[0]), int(out[1]), int(out[2]))

    def needs_adjustment(self, detected_count: int) -> bool:           # :65-67
        return detected_count < self.min_speakers or detected_count > self.max_speakers

    def target_count(self, detected_count: int) -> int:                # :70-76
        return min(max(detected_count, self.min_speakers), self.max_speakers)
