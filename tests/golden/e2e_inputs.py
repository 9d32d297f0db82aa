"""Seeded input of BASELINE configs[4]: one synthetic recording of `hours` hours as the clustering stage
sees it (OfflineDiarizerManager.swift:270-467): 3 local speaker slots per 2 s step (OfflineDiarizerTypes.swift:46-55) ->
N = 5 400 x hours embeddings, 256-d fp32, 128-d PLDA features fp64, chunk index per embedding, synthetic positive Phi.

Same construction as round 2's bench generator (speaker centres + 0.03 N(0, I), 3 distinct speakers per step), but every
floating-point step has ONE IEEE rounding in a fixed order (sequential cumsum for the centre norms, element-wise products and
sums otherwise), so the bytes do not depend on numpy's SIMD reduction order; `input_sha256` in the committed digest lets the
GPU test and bench.py check that they regenerated the same bytes before they compare results.

Used by tests/golden/make_e2e_digest.py (CPU: oracle + the reference's linkage build), tests/test_gpu_e2e_digest.py and
bench.py (device).  Not part of the product and not part of oracle/.
"""
import hashlib

import numpy as np


def e2e_session(hours: float = 8.0, speakers: int = 12, seed: int = 5, sigma: float = 0.03):
    rng = np.random.default_rng(seed)
    n_win = int(hours * 3600 / 2)
    n = 3 * n_win
    centers = rng.standard_normal((speakers, 256))
    norm = np.cumsum(centers * centers, axis=1)[:, -1]
    centers = centers * (1.0 / np.sqrt(norm))[:, None]
    spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_win)]).reshape(-1)
    emb = (centers[spk] + sigma * rng.standard_normal((n, 256))).astype(np.float32)
    phi = np.linspace(2.0, 1.0, 128)
    rho = (rng.standard_normal((speakers, 128)) * np.sqrt(phi))[spk] + rng.standard_normal((n, 128))
    chunks = np.repeat(np.arange(n_win), 3).astype(np.int32)
    return {"emb": np.ascontiguousarray(emb), "rho": np.ascontiguousarray(rho), "chunks": chunks, "phi": phi, "spk": spk,
            "speakers": speakers, "hours": hours}


def sha256(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def input_digest(s) -> str:
    h = hashlib.sha256()
    for k in ("emb", "rho", "chunks", "phi"):
        h.update(np.ascontiguousarray(s[k]).tobytes())
    return h.hexdigest()


def round9(a) -> np.ndarray:
    """Centroids are compared at 1e-9 (device VBx sums in a different order than the CPU restatement: gamma agrees to ~1e-12);
    the committed file holds the fp64 values themselves, the digest is informational."""
    return np.round(np.asarray(a, np.float64), 9) + 0.0
