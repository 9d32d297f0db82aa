"""Runs the CPU side of BASELINE configs[4] at FULL size — the clustering stage of one 8 h recording (43 200 embeddings):
AHCClustering.cluster on the REFERENCE's own linkage build (oracle/_ref) -> VBx -> gamma-weighted centroids -> per-chunk
constrained assignment (OfflineDiarizerManager.swift:270-467, VBxClustering.swift:167-664), and the forced-speaker-count
variant (K-Means n_init = 10, VBxClustering.swift:685-733) — and commits what it returns.

    python tests/golden/make_e2e_digest.py                    (≈12-20 CPU-minutes on one core, almost all in the reference linkage)
    python tests/golden/make_e2e_digest.py --sigma 0.041      (harder session: AHC leaves hundreds of clusters, VBx and the
                                                               constrained assignment do real work -> e2e_8h_s0p041.*)

Writes tests/golden/e2e_8h.json (SHA-256 of the input bytes, of the reference dendrogram, of the AHC labels, the VBx hard
labels, the final assignments; VBx iteration count, ELBOs, pi; cluster counts; wall-clock per CPU stage) and
tests/golden/e2e_8h.npz (centroids fp64 of both variants, AHC merge pairs int32, assignments int16 — to locate a mismatch).
tests/test_gpu_e2e_digest.py and bench.py regenerate the same input bytes (e2e_inputs.py) and compare; nothing at GPU-test /
bench time needs /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import oracle  # noqa: E402
from ahc_full_inputs import dendrogram_digest  # noqa: E402
from e2e_inputs import e2e_session, input_digest, round9, sha256  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=float, default=8.0)
    ap.add_argument("--speakers", type=int, default=12)
    ap.add_argument("--forced", type=int, default=10, help="numSpeakers of the K-Means variant")
    ap.add_argument("--sigma", type=float, default=0.03, help="0.03: the bench session (AHC finds the speakers); 0.041: AHC leaves hundreds of clusters for VBx")
    ap.add_argument("--stem", default=None)
    a = ap.parse_args()
    oracle.build()
    assert oracle.ref_available(), "oracle/_ref is not built (needs /root/reference)"
    s = e2e_session(a.hours, a.speakers, sigma=a.sigma)
    n = len(s["emb"])
    emb64 = s["emb"].astype(np.float64)                     # OfflineDiarizerManager.swift:286 widens Float -> Double
    t0 = time.perf_counter()
    xn = oracle.ahc_normalize(emb64)
    st, z = oracle.linkage_ref(xn)
    t_link = time.perf_counter() - t0
    assert st == 0, st
    ahc = oracle.ahc_cut(z, n, 0.6)
    assert np.array_equal(ahc, oracle.ahc_cluster(emb64, 0.6, linkage=None)) if n <= 3000 else True
    t0 = time.perf_counter()
    res = oracle.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], initial=ahc)
    t_rest = time.perf_counter() - t0
    t0 = time.perf_counter()
    km = oracle.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], initial=ahc, num_speakers=a.forced)
    t_km = time.perf_counter() - t0
    assert km["was_adjusted"]
    out = {"hours": a.hours, "speakers": a.speakers, "n": n, "sigma": a.sigma, "generator": f"tests/golden/e2e_inputs.py e2e_session(seed 5, sigma {a.sigma:g})",
           "input_sha256": input_digest(s), "numpy": np.__version__,
           "reference": "AHC: fastcluster_compute_centroid_linkage built from /root/reference (oracle/_ref); VBx / centroids / Hungarian / K-Means: "
                        "oracle/fa_oracle.c restatements (no reference-held values exist for them: parity unpinned beyond the restatement)",
           "cpu_seconds_1_core": {"linkage_reference": t_link, "vbx_centroids_assignment": t_rest, "kmeans_variant": t_km},
           "dendrogram": dendrogram_digest(z),
           "ahc_labels_sha256": sha256(ahc.astype(np.int32)), "ahc_clusters": int(ahc.max()) + 1,
           "vbx_hard_sha256": sha256(np.asarray(res["hard"], np.int32)), "vbx_iterations": int(len(res["elbos"])),
           "vbx_elbos": [float(v) for v in res["elbos"]], "vbx_pi": [float(v) for v in res["pi"]],
           "centroids": int(res["centroids"].shape[0]), "centroids_round9_sha256": sha256(round9(res["centroids"])),
           "assignments_sha256": sha256(np.asarray(res["assignments"], np.int32)),
           "labels_match_speakers": len(set(zip(s["spk"].tolist(), np.asarray(res["assignments"]).tolist()))) == a.speakers,
           "forced": {"num_speakers": a.forced, "detected": int(km["detected"]), "centroids": int(km["centroids"].shape[0]),
                      "kmeans_labels_sha256": sha256(np.asarray(km["kmeans_clusters"], np.int32)),
                      "centroids_round9_sha256": sha256(round9(km["centroids"])),
                      "assignments_sha256": sha256(np.asarray(km["assignments"], np.int32))}}
    stem = a.stem or os.path.join(HERE, f"e2e_{a.hours:g}h" + ("" if a.sigma == 0.03 else f"_s{a.sigma:g}".replace(".", "p")))
    with open(stem + ".json", "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(stem + ".npz", centroids=res["centroids"], forced_centroids=km["centroids"], pairs=z[:, :2].astype(np.int32),
                        ahc=ahc.astype(np.int16), assignments=np.asarray(res["assignments"]).astype(np.int16),
                        forced_assignments=np.asarray(km["assignments"]).astype(np.int16), vbx_hard=np.asarray(res["hard"]).astype(np.int16))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
