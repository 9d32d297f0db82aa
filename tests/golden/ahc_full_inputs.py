"""Seeded inputs of BASELINE configs[2] that reproduce bit-for-bit on any box with this numpy.

Both distributions use numpy's PCG64 stream (`default_rng(seed)`), fp64, and a normalisation made only of
operations with one IEEE rounding each in a fixed order (sequential `cumsum` for the squared norm, then `1/sqrt`, then a
product), so the bytes of X do not depend on numpy's SIMD reduction order.  `input_sha256` in the committed digest files
lets every consumer (tests, bench) check that it regenerated the same bytes before it compares dendrograms.

Used by tests/golden/make_ahc_full_digest.py (reference run, CPU), tests/test_gpu_ahc.py and bench.py (device run).
Not part of the product and not part of oracle/.
"""
import hashlib

import numpy as np

THRESHOLDS = (0.6, 1.0, 1.05, 1.2)


def _unit_rows(x: np.ndarray) -> np.ndarray:
    norm = np.cumsum(x * x, axis=1)[:, -1]
    return np.ascontiguousarray(x * (1.0 / np.sqrt(norm))[:, None])


def ahc_input(dist: str, n: int, d: int = 256, seed: int = 0) -> np.ndarray:
    """dist 'iid': N(0,1) rows, unit L2.  dist 'mix': K=64 speakers, row i = unit(c[i mod 64] + 0.02 eps)."""
    rng = np.random.default_rng(seed)
    if dist == "iid":
        return _unit_rows(rng.standard_normal((n, d)))
    if dist == "mix":
        c = _unit_rows(rng.standard_normal((64, d)))
        eps = rng.standard_normal((n, d))
        return _unit_rows(c[np.arange(n) % 64] + 0.02 * eps)
    raise ValueError(dist)


TIED_KINDS = ("dup30", "silence5", "grid64")


def ahc_tied_input(kind: str, hours: float = 8.0, seed: int = 1) -> np.ndarray:
    """Inputs WITH exact ties at the size of BASELINE configs[4] (5 400 x hours rows, 256-d): the rows of the e2e session
    (e2e_inputs.e2e_session: 12 speakers, 3 per 2 s step), widened to fp64 and unit-normalised like AHCClustering.swift:70-105, then
      dup30    — 30 % of the rows overwritten by copies of other rows (zero distances, repeated embeddings of a long turn),
      silence5 — 5 % of the rows replaced by ONE row ("digital silence": 2 160 identical embeddings, a 2 160-way tie at distance 0),
      grid64   — every coordinate rounded to the 1/64 grid and NOT re-normalised: squared distances are multiples of 2^-12, so
                 DIFFERENT pairs tie exactly at non-zero distances in nearly every scan (embeddings from a quantised model).
    The reference's order among exact ties is its heap layout (fastcluster_internal.hpp:1685-1799); the digests of what it
    returns on these inputs are what the device's tie route is held to (make_ahc_full_digest.py --tied)."""
    from e2e_inputs import e2e_session
    x = _unit_rows(e2e_session(hours, 12, seed=5)["emb"].astype(np.float64))
    n = len(x)
    rng = np.random.default_rng(seed)
    if kind == "tie_free":
        return x
    if kind == "dup30":
        k = int(0.3 * n)
        dst, src = rng.integers(0, n, k), rng.integers(0, n, k)
        x[dst] = x[src]
        return np.ascontiguousarray(x)
    if kind == "silence5":
        x[rng.permutation(n)[: n // 20]] = x[0]
        return np.ascontiguousarray(x)
    if kind == "grid64":
        return np.ascontiguousarray(np.round(x * 64.0) / 64.0)
    raise ValueError(kind)


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dendrogram_digest(z: np.ndarray) -> dict:
    """Digest of a SciPy-style [(N-1),4] fp64 dendrogram: the bytes as the C ABI writes them, and separately the merge
    pairs (int32) and the heights, so a mismatch can be located."""
    z = np.ascontiguousarray(z, np.float64)
    return {"dendrogram_sha256": sha256(z), "pairs_sha256": sha256(z[:, :2].astype(np.int32)),
            "heights_sha256": sha256(z[:, 2]), "sizes_sha256": sha256(z[:, 3].astype(np.int32)),
            "height_sum": float(np.sum(z[:, 2])), "height_max": float(z[:, 2].max()),
            "height_inversions": int((np.diff(z[:, 2]) < 0).sum())}
