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
