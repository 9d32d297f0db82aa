"""BASELINE configs[4] at FULL size on the device: the clustering stage of one 8 h recording (43 200 embeddings) through ONE library
call (fa_offline_cluster_ex) against the CPU side committed in tests/golden/e2e_8h*.json / .npz by make_e2e_digest.py — AHC on the
REFERENCE's own linkage build (oracle/_ref: ≈12 CPU-minutes, zero GPU budget), VBx / centroids / per-chunk Hungarian / K-Means from
the C restatements (reference: OfflineDiarizerManager.swift:270-467, VBxClustering.swift:167-664,685-733).

Bars: AHC labels, VBx hard labels, iteration count and final assignments bit-exact (SHA-256 of the int32 vectors); ELBOs and
centroids within 1e-9 (the device sums gamma^T rho in a different order than the sequential CPU loops).  Two sessions: sigma 0.03
(the bench session: AHC already finds the 12 speakers) and sigma 0.041 (AHC leaves hundreds of clusters: VBx prunes them to 12 and
the constrained assignment moves thousands of embeddings)."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)


def load(stem):
    jp = os.path.join(GOLD, stem + ".json")
    if not os.path.exists(jp):
        pytest.skip(f"{stem}.json not committed")
    with open(jp) as f:
        gold = json.load(f)
    from e2e_inputs import e2e_session, input_digest
    s = e2e_session(gold["hours"], gold["speakers"], sigma=gold.get("sigma", 0.03))
    assert input_digest(s) == gold["input_sha256"], "this numpy regenerates different input bytes: the digests do not apply"
    return gold, np.load(os.path.join(GOLD, stem + ".npz")), s


@pytest.mark.parametrize("stem", ["e2e_8h", "e2e_8h_s0p041"])
def test_cluster_stage_8h_equals_cpu_digests(fa, gpu_ctx, stem):
    from e2e_inputs import sha256
    gold, aux, s = load(stem)
    res = fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=gpu_ctx, intermediates=True)
    ahc = np.asarray(res.initial_clusters, np.int32)
    where = np.nonzero(ahc != aux["ahc"].astype(np.int32))[0]
    assert sha256(ahc) == gold["ahc_labels_sha256"], f"AHC labels differ at {where[:5]} ({where.size} rows); stats {res.info['ahc']}"
    assert res.info["initial_clusters"] == gold["ahc_clusters"]
    assert res.info["vbx_iterations"] == gold["vbx_iterations"]
    np.testing.assert_allclose(res.info["elbos"], gold["vbx_elbos"], rtol=1e-9, atol=0)
    hard = np.asarray(res.info["vbx_hard"], np.int32)
    assert sha256(hard) == gold["vbx_hard_sha256"], int((hard != aux["vbx_hard"]).sum())
    assert res.centroids.shape == aux["centroids"].shape
    np.testing.assert_allclose(res.centroids, aux["centroids"], rtol=0, atol=1e-9)
    got = np.asarray(res.assignments, np.int32)
    assert sha256(got) == gold["assignments_sha256"], int((got != aux["assignments"]).sum())
    assert res.info["ahc"]["merges"] == gold["n"] - 1


def test_cluster_stage_8h_forced_speaker_count(fa, gpu_ctx):
    """numSpeakers forces the K-Means n_init = 10 fallback (VBxClustering.swift:685-733) on all 43 200 training rows."""
    from e2e_inputs import sha256
    gold, aux, s = load("e2e_8h")
    cfg = fa.OfflineClusteringConfig(num_speakers=gold["forced"]["num_speakers"])
    res = fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], cfg, ctx=gpu_ctx, intermediates=True)
    assert res.info["was_adjusted"] == 1 and res.centroids.shape[0] == gold["forced"]["centroids"]
    np.testing.assert_allclose(res.centroids, aux["forced_centroids"], rtol=0, atol=1e-9)
    got = np.asarray(res.assignments, np.int32)
    assert sha256(got) == gold["forced"]["assignments_sha256"], int((got != aux["forced_assignments"]).sum())
