"""K-Means fallback on the device vs the CPU oracle: labels, centroids, inertias and the winning run bit-exact (same draws,
same summation orders), plus the speaker-count-constrained clustering stage end to end."""
import numpy as np
import pytest
from test_gpu_pipeline import synth_session
from test_oracle_kmeans import SIX

pytestmark = pytest.mark.gpu


def test_reference_cases(fa, gpu_ctx, oracle_mod):        # KMeansClusteringTests.swift:10-131
    K = fa.KMeansClustering
    lab = K.cluster(SIX, 3, 100, 42, ctx=gpu_ctx)
    assert len(lab) == 6 and len(set(lab)) == 3 and lab == oracle_mod.kmeans(SIX, 3, 100, 42)[0].tolist()
    assert K.cluster([[1.0, 0.0], [1.1, 0.1], [0.9, 0.2]], 1, 100, 42, ctx=gpu_ctx) == [0, 0, 0]
    lab, cen = K.cluster_with_centroids([[1.0, 0.0], [0.0, 1.0]], 5, 100, 42, ctx=gpu_ctx)
    assert lab == [0, 1] and np.array_equal(cen, [[1.0, 0.0], [0.0, 1.0]])
    lab, cen = K.cluster_with_centroids([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [0.0, 1.0]], 2, 100, 42, ctx=gpu_ctx)
    assert cen.shape == (2, 2) and len(lab) == 4
    assert K.cluster(SIX, 3, seed=12345, ctx=gpu_ctx) == K.cluster(SIX, 3, seed=12345, ctx=gpu_ctx)
    rng = oracle_mod.SeededRNG(42)
    emb = [[rng.random_double(-1.0, 1.0) for _ in range(192)] for _ in range(20)]
    lab = K.cluster(emb, 3, 100, 42, ctx=gpu_ctx)
    assert len(set(lab)) == 3 and lab == oracle_mod.kmeans(emb, 3, 100, 42)[0].tolist()
    # guards (:46-56)
    assert K.cluster(np.zeros((0, 4)), 3, ctx=gpu_ctx) == []
    assert K.cluster(np.zeros((3, 0)), 2, ctx=gpu_ctx) == [0, 0, 0]
    assert K.cluster(SIX, 0, ctx=gpu_ctx) == [0] * 6


@pytest.mark.parametrize("n,d,k,speakers,noise,seed", [(600, 256, 4, 4, 0.05, 0), (2000, 256, 12, 9, 0.08, 1), (777, 33, 20, 5, 0.3, 2),
                                                     (1500, 64, 3, 7, 0.5, 3), (300, 8, 40, 3, 0.2, 4)])
def test_single_run_matches_oracle(fa, gpu_ctx, oracle_mod, n, d, k, speakers, noise, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((speakers, d))
    x = c[rng.integers(0, speakers, n)] + noise * rng.standard_normal((n, d))
    for s in (seed, seed + 100):
        lab_r, cen_r, it_r = oracle_mod.kmeans(x, k, 100, s)
        lab_g, cen_g = fa.KMeansClustering.cluster_with_centroids(x, k, 100, s, ctx=gpu_ctx)
        assert lab_g == lab_r.tolist()
        np.testing.assert_array_equal(cen_g, cen_r)


def test_empty_clusters_and_duplicates(fa, gpu_ctx, oracle_mod):
    x = np.repeat(np.eye(3), 5, axis=0)                   # 3 distinct points, k = 3 and 5: empty clusters get re-seeded
    for k in (3, 5):
        for seed in range(6):
            lab_r, cen_r, _ = oracle_mod.kmeans(x, k, 50, seed)
            lab_g, cen_g = fa.KMeansClustering.cluster_with_centroids(x, k, 50, seed, ctx=gpu_ctx)
            assert lab_g == lab_r.tolist()
            np.testing.assert_array_equal(cen_g, cen_r)
    z = np.zeros((10, 4)); z[3] = [np.nan, 0, 0, 1]        # zero rows stay unnormalised (:135), NaN distances never win
    z[7] = [2.0, 0, 0, 0]
    lab_r, cen_r, _ = oracle_mod.kmeans(z, 2, 20, 1)
    lab_g, cen_g = fa.KMeansClustering.cluster_with_centroids(z, 2, 20, 1, ctx=gpu_ctx)
    assert lab_g == lab_r.tolist()
    np.testing.assert_array_equal(cen_g, cen_r)


@pytest.mark.parametrize("n,k,seed", [(900, 5, 0), (2500, 8, 1)])
def test_n_init_matches_oracle(fa, gpu_ctx, oracle_mod, n, k, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((k + 2, 256))
    x = c[rng.integers(0, k + 2, n)] + 0.4 * rng.standard_normal((n, 256))
    lab_r, cen_r, best_r, inert_r = oracle_mod.kmeans_ninit(x, k, 100, 10, 0)
    det = {}
    lab_g, cen_g = fa.KMeansClustering.cluster_with_centroids_n_init(x, k, 100, 10, 0, ctx=gpu_ctx, details=det)
    np.testing.assert_array_equal(det["inertias"], inert_r)
    assert det["best_run"] == best_r and lab_g == lab_r.tolist()
    np.testing.assert_array_equal(cen_g, cen_r)
    # the n_init guard: a single run with the base seed
    lab1, _ = fa.KMeansClustering.cluster_with_centroids_n_init(x, k, 100, 1, 7, ctx=gpu_ctx)
    assert lab1 == oracle_mod.kmeans(x, k, 100, 7)[0].tolist()


@pytest.mark.parametrize("forced", [2, 6])
def test_constrained_stage_matches_cpu_restatement(fa, gpu_ctx, oracle_mod, forced):
    emb, rho, chunks, phi, spk = synth_session(300, 4, 0)
    cfg = fa.OfflineClusteringConfig(num_speakers=forced)
    res = fa.cluster_embeddings_stagewise(emb, rho, chunks, phi, cfg, ctx=gpu_ctx)   # exposes the K-Means intermediates
    one = fa.cluster_embeddings(emb, rho, chunks, phi, cfg, ctx=gpu_ctx)              # the single device-resident call
    assert one.assignments == res.assignments and np.array_equal(one.centroids, res.centroids) and one.info["was_adjusted"] == 1
    ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi, num_speakers=forced)
    assert res.vbx.was_adjusted and ref["was_adjusted"] and res.vbx.original_cluster_count == ref["detected"] == 4
    assert res.vbx.hard_clusters[0] == ref["kmeans_clusters"].tolist()
    np.testing.assert_array_equal(res.centroids, ref["centroids"])
    assert res.assignments == ref["assignments"].tolist() and len(set(res.assignments)) == forced
    # within bounds: nothing is adjusted and the constrained assignment stays on
    cfg = fa.OfflineClusteringConfig(min_speakers=2, max_speakers=6)
    res = fa.cluster_embeddings(emb, rho, chunks, phi, cfg, ctx=gpu_ctx)
    assert not res.info["was_adjusted"] and res.assignments == oracle_mod.cluster_embeddings(emb, rho, chunks, phi)["assignments"].tolist()
