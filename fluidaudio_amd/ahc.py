"""Host-side mirror of ``AHCClustering`` (reference:
Sources/FluidAudio/Diarizer/Offline/Clustering/AHCClustering.swift:12-211) and of the
``fastcluster_compute_centroid_linkage`` FFI it calls
(Sources/FastClusterWrapper/include/FastClusterWrapper.h:35-41), over the HIP C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def linkage(data, mode: int = L.AHC_MODE_AUTO, ctx: L.Context | None = None, return_stats: bool = False):
    """Centroid-linkage dendrogram (SciPy format, merge order) of row-major fp64 `data` [N, d].

    Returns (status, Z[(N-1), 4]) like the C ABI: a non-zero status leaves Z unspecified."""
    ctx = ctx or L.default_context()
    x = np.ascontiguousarray(data, np.float64)
    n, d = x.shape
    z = np.zeros((max(n - 1, 0), 4), np.float64)
    stats = L.AhcStats()
    st = L.lib().fa_ahc_linkage(ctx.handle, x.ctypes.data, n, d, z.ctypes.data, z.size, mode, 0, C.byref(stats))
    return (st, z, stats.as_dict()) if return_stats else (st, z)


def linkage_batch(problems, mode: int = L.AHC_MODE_AUTO, ctx: L.Context | None = None, return_stats: bool = False):
    """Centroid-linkage dendrograms of several independent problems ([N_k, d] fp64 each, same d) in one call
    (fa_ahc_linkage_batch: the merge chains advance together).  Returns (statuses list, [Z_k]) (+ stats dicts)."""
    ctx = ctx or L.default_context()
    xs = [np.ascontiguousarray(p, np.float64) for p in problems]
    k = len(xs)
    if k == 0:
        return ([], [], []) if return_stats else ([], [])
    d = xs[0].shape[1]
    assert all(x.ndim == 2 and x.shape[1] == d for x in xs)
    zs = [np.zeros((max(x.shape[0] - 1, 0), 4), np.float64) for x in xs]
    dp = (C.c_void_p * k)(*[x.ctypes.data for x in xs])
    dummy = np.zeros(4)   # the reference contract rejects a NULL output pointer even when nothing is written (n <= 1)
    zp = (C.c_void_p * k)(*[z.ctypes.data if z.size else dummy.ctypes.data for z in zs])
    dp = (C.c_void_p * k)(*[x.ctypes.data if x.size else dummy.ctypes.data for x in xs])
    ns = (C.c_size_t * k)(*[x.shape[0] for x in xs])
    st = (C.c_int32 * k)()
    stats = (L.AhcStats * k)()
    L.lib().fa_ahc_linkage_batch(ctx.handle, k, dp, ns, d, zp, mode, 0, stats, st)
    out = ([int(v) for v in st], zs)
    return out + ([s.as_dict() for s in stats],) if return_stats else out


def fastcluster_compute_centroid_linkage(data) -> tuple[int, np.ndarray]:
    """The exact reference symbol (no context argument, library-owned default context)."""
    x = np.ascontiguousarray(data, np.float64)
    n, d = x.shape
    z = np.zeros((max(n - 1, 0), 4), np.float64)
    st = L.lib().fastcluster_compute_centroid_linkage(x.ctypes.data, n, d, z.ctypes.data, z.size)
    return st, z


def check_dendrogram(z, n: int) -> None:
    """A linkage matrix the cut can walk: (n - 1) rows; the children of row r are two different nodes that exist when it is formed
    (leaves 0 .. n-1, merges n .. n+r-1) and no node is merged twice.  The reference only ever cuts what its own wrapper wrote
    (AHCClustering.swift:40-58) and has no such check; a C caller can hand over anything, and a child index >= its own row's node
    would send the walk out of bounds or around a cycle."""
    z = np.asarray(z, np.float64)
    if n <= 1:
        return
    if z.shape != (n - 1, 4):
        raise ValueError(f"dendrogram of {n} points needs {n - 1} rows of 4 values, got shape {z.shape}")
    kids = z[:, :2]
    if not np.isfinite(kids).all() or (kids != np.floor(kids)).any():
        raise ValueError("dendrogram: child indices must be whole numbers")
    limit = (n + np.arange(n - 1))[:, None]
    if (kids < 0).any() or (kids >= limit).any() or (kids[:, 0] == kids[:, 1]).any():
        raise ValueError("dendrogram: a row merges a node that does not exist yet, or a node with itself")
    if len(np.unique(kids)) != kids.size:
        raise ValueError("dendrogram: a node is merged twice")


def cut(z, n: int, threshold: float) -> np.ndarray:
    """assignmentsFromDendrogram + remapClusterIds (:124-210) with the threshold clamp (:112-121)."""
    z = np.ascontiguousarray(z, np.float64)
    check_dendrogram(z.reshape(-1, 4) if z.size else z, n)
    labels = np.zeros(max(n, 1), np.int32)
    st = L.lib().fa_ahc_cut(z.ctypes.data if z.size else None, n, float(threshold), labels.ctypes.data)
    if st != L.SUCCESS:
        raise L.FluidAudioHipError(st, "fa_ahc_cut")
    return labels[:n]


class AHCClustering:
    """struct AHCClustering (:12)."""

    def __init__(self, ctx: L.Context | None = None, mode: int = L.AHC_MODE_AUTO):
        self._ctx, self.mode = ctx, mode
        self.last_stats: dict | None = None
        self.last_status: int = L.SUCCESS

    def cluster(self, embedding_features, threshold: float) -> list[int]:
        """cluster(embeddingFeatures:threshold:) (:20-67)."""
        count = len(embedding_features)
        if count == 0:
            return []
        first = embedding_features[0]
        dim = len(first)
        if dim == 0:
            return [0] * count
        if count == 1:
            return [0]
        x = np.ascontiguousarray(embedding_features, np.float64)
        ctx = self._ctx or L.default_context()
        labels = np.zeros(count, np.int32)
        stats = L.AhcStats()
        st = L.lib().fa_ahc_cluster(ctx.handle, x.ctypes.data, count, dim, float(threshold), self.mode,
                                    labels.ctypes.data, C.byref(stats))
        self.last_status, self.last_stats = st, stats.as_dict()
        # on failure the library has already filled labels with 0..<count (:52-55)
        return np.asarray(labels).tolist()   # (one C loop: a Python-level comprehension over 43 200 labels costs 2-3 ms)
