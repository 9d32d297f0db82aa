"""Host-side mirror of what ``OfflineDiarizerManager.cluster`` does with the VBx posteriors (reference:
Sources/FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift): ``computeCentroids`` (:613-691) and
``assignEmbeddings`` (:789-822), over the HIP C ABI (csrc/post.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def compute_centroids(embedding_features, gamma, pi, ctx: L.Context | None = None):
    """gamma-weighted centroids of the speakers with pi > 1e-7.  Returns (centroids [K, d], map [S] -> row or -1)."""
    emb = np.ascontiguousarray(embedding_features, np.float64)
    gamma = np.ascontiguousarray(gamma, np.float64)
    pi = np.ascontiguousarray(pi, np.float64)
    n, d = emb.shape if emb.ndim == 2 else (0, 0)
    S = pi.size
    if n == 0 or d == 0 or S == 0:
        return np.zeros((0, d), np.float64), np.full(S, -1, np.int32)  # guards (:618-628)
    ctx = ctx or L.default_context()
    cent = np.zeros((S, d), np.float64)
    mp = np.zeros(S, np.int32)
    k = C.c_int32()
    ctx.check(L.lib().fa_vbx_weighted_centroids(ctx.handle, emb.ctypes.data, n, d, gamma.ctypes.data, pi.ctypes.data, S,
                                                cent.ctypes.data, mp.ctypes.data, C.byref(k)), "fa_vbx_weighted_centroids")
    return cent[:k.value], mp


def assign_embeddings(embedding_features, centroids, ctx: L.Context | None = None) -> list:
    emb = np.ascontiguousarray(embedding_features, np.float64)
    if emb.size == 0:
        return []  # :794
    cen = np.ascontiguousarray(centroids, np.float64)
    n, d = emb.shape
    K = cen.shape[0] if cen.ndim == 2 else 0
    if K == 0:
        return [0] * n  # :795-797
    ctx = ctx or L.default_context()
    out = np.zeros(n, np.int32)
    ctx.check(L.lib().fa_assign_cosine(ctx.handle, emb.ctypes.data, n, d, cen.ctypes.data, K, out.ctypes.data), "fa_assign_cosine")
    return np.asarray(out).tolist()


def centroid_scores(embedding_features, centroids, ctx: L.Context | None = None) -> np.ndarray:
    """centroidScores (:789-798): [n, K] cosine scores."""
    emb = np.ascontiguousarray(embedding_features, np.float64)
    cen = np.ascontiguousarray(centroids, np.float64)
    n = emb.shape[0] if emb.ndim == 2 else 0
    K = cen.shape[0] if cen.ndim == 2 else 0
    out = np.zeros((n, K), np.float64)
    if n and K:
        ctx = ctx or L.default_context()
        ctx.check(L.lib().fa_centroid_scores(ctx.handle, emb.ctypes.data, n, emb.shape[1], cen.ctypes.data, K, out.ctypes.data),
                  "fa_centroid_scores")
    return out


class ConstrainedClusterAssignment:
    """Mirror of ``ConstrainedClusterAssignment`` (reference:
    Sources/FluidAudio/Diarizer/Offline/Clustering/ConstrainedClusterAssignment.swift:20-42)."""

    @staticmethod
    def assign(scores, chunk_indices, ctx: L.Context | None = None) -> list:
        n = len(chunk_indices)
        if len(scores) != n:
            raise ValueError("scores and chunkIndices must be parallel arrays")  # precondition (:21-24)
        if n == 0:
            return []
        sc = np.ascontiguousarray(scores, np.float64)
        K = sc.shape[1] if sc.ndim == 2 else 0
        ch = np.ascontiguousarray(chunk_indices, np.int32)
        out = np.zeros(n, np.int32)
        ctx = ctx or L.default_context()
        ctx.check(L.lib().fa_constrained_assign(ctx.handle, sc.ctypes.data if K else None, n, K, ch.ctypes.data, out.ctypes.data),
                  "fa_constrained_assign")
        return np.asarray(out).tolist()


class HungarianAssignment:
    """``HungarianAssignment.maxScoreAssignment`` (reference: Sources/FluidAudio/Diarizer/HungarianAssignment.swift:67-97)."""

    @staticmethod
    def max_score_assignment(scores, ctx: L.Context | None = None) -> list:
        rows = len(scores)
        if rows == 0:
            return []
        res = ConstrainedClusterAssignment.assign(scores, [0] * rows, ctx=ctx)
        return [v if v >= 0 else -1 for v in res]
