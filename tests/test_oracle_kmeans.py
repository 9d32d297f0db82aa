"""Oracle pins for the K-Means fallback and SpeakerCountConstraints: the cases of the reference's own tests
(Tests/FluidAudioTests/Diarizer/Clustering/KMeansClusteringTests.swift:10-131,
Tests/FluidAudioTests/Diarizer/Offline/SpeakerCountConstraintsTests.swift:10-136).  The reference tests hold structural
answers only (cluster counts, determinism), so the draw sequence itself stays unpinned against a Swift toolchain."""
import numpy as np
import pytest

import oracle

SIX = [[1.0, 0.0], [1.1, 0.1], [0.0, 1.0], [0.1, 1.1], [-1.0, 0.0], [-0.9, 0.1]]


def test_requested_cluster_count():                      # testKMeansProducesRequestedClusterCount (:10-32)
    lab, cen, _ = oracle.kmeans(SIX, 3, 100, 42)
    assert len(lab) == 6 and len(set(lab.tolist())) == 3 and cen.shape == (3, 2)
    assert lab[0] == lab[1] and lab[2] == lab[3] and lab[4] == lab[5]


def test_single_cluster():                                # testKMeansHandlesSingleCluster (:34-49)
    lab, _, _ = oracle.kmeans([[1.0, 0.0], [1.1, 0.1], [0.9, 0.2]], 1, 100, 42)
    assert lab.tolist() == [0, 0, 0]


def test_more_clusters_than_embeddings():                 # :51-66
    lab, cen, _ = oracle.kmeans([[1.0, 0.0], [0.0, 1.0]], 5, 100, 42)
    assert lab.tolist() == [0, 1]
    assert np.array_equal(cen, [[1.0, 0.0], [0.0, 1.0]])  # raw embeddings come back as the centroids (:57-59)


def test_centroids_returned():                            # testKMeansComputesCentroids (:68-86)
    lab, cen, _ = oracle.kmeans([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [0.0, 1.0]], 2, 100, 42)
    assert cen.shape == (2, 2) and len(lab) == 4 and lab[0] == lab[1] != lab[2] == lab[3]


def test_deterministic_with_seed():                       # :90-109
    a = oracle.kmeans(SIX, 3, 300, 12345)[0]
    b = oracle.kmeans(SIX, 3, 300, 12345)[0]
    assert np.array_equal(a, b)


def test_realistic_dimension():                           # testKMeansWithRealisticEmbeddingDimension (:113-130)
    rng = oracle.SeededRNG(42)
    emb = [[rng.random_double(-1.0, 1.0) for _ in range(192)] for _ in range(20)]
    assert all(-1.0 <= v <= 1.0 for row in emb for v in row)
    lab, _, _ = oracle.kmeans(emb, 3, 100, 42)
    assert len(lab) == 20 and len(set(lab.tolist())) == 3


def test_degenerate_guards():                             # :46-56
    assert oracle.kmeans(np.zeros((0, 4)), 3)[0].size == 0
    assert oracle.kmeans(np.zeros((3, 0)), 2)[0].tolist() == [0, 0, 0]
    assert oracle.kmeans(SIX, 0)[0].tolist() == [0] * 6
    assert oracle.kmeans(SIX, -2)[0].tolist() == [0] * 6


def test_lcg_and_bounded_draw():
    r = oracle.SeededRNG(0)
    assert r.next() == 1442695040888963407                # state * a + c with state 0 (:219-222)
    assert r.next() == (1442695040888963407 * 6364136223846793005 + 1442695040888963407) % 2 ** 64
    r = oracle.SeededRNG(7)
    draws = [r.next_upper_bound(10) for _ in range(2000)]
    assert min(draws) == 0 and max(draws) == 9
    # Lemire's method is the high word of the 128-bit product unless the low word falls in the rejection zone
    r1, r2 = oracle.SeededRNG(99), oracle.SeededRNG(99)
    for _ in range(100):
        assert r1.next_upper_bound(1 << 32) == (r2.next() * (1 << 32)) >> 64
    p = oracle.SeededRNG(3).shuffled_indices(50)
    assert sorted(p.tolist()) == list(range(50)) and p.tolist() != list(range(50))


def test_ninit_picks_lowest_inertia_first_on_ties():      # clusterWithCentroidsNInit (:99-129)
    lab, cen, best, inert = oracle.kmeans_ninit(SIX, 3, 100, 10, 0)
    singles = [oracle.kmeans(SIX, 3, 100, s) for s in range(10)]
    assert np.isfinite(inert).all()
    assert best == int(np.argmin(inert)) and inert[best] == inert.min()
    assert np.array_equal(lab, singles[best][0]) and np.array_equal(cen, singles[best][1])
    # guard: n <= numClusters or nInit <= 1 -> the single seeded run
    lab1, _, b1, _ = oracle.kmeans_ninit(SIX, 3, 100, 1, 5)
    assert b1 == 0 and np.array_equal(lab1, oracle.kmeans(SIX, 3, 100, 5)[0])
    assert oracle.kmeans_ninit(SIX, 6, 100, 10, 0)[0].tolist() == [0, 1, 2, 3, 4, 5]


def test_empty_cluster_reseed_keeps_k():
    # duplicates force an empty cluster on the first update for some seeds; the run must still finish with valid labels
    x = np.repeat(np.eye(3), 5, axis=0)
    for seed in range(8):
        lab, cen, it = oracle.kmeans(x, 3, 50, seed)
        assert lab.min() >= 0 and lab.max() <= 2 and cen.shape == (3, 3) and 1 <= it <= 50


@pytest.mark.parametrize("args,expect", [
    ((100, None, None, None), (None, 1, 100)),            # :10-20
    ((100, 3, 1, 10), (3, 3, 3)),                         # :22-32
    ((5, None, 2, 20), (None, 2, 5)),                     # :34-43
    ((100, None, 10, 5), (5, 5, 5)),                      # :47-56
    ((100, 0, None, None), (1, 1, 1)),                    # :60-69
    ((100, -5, None, None), (1, 1, 1)),                   # :71-80
    ((100, None, 0, 5), (None, 1, 5)),                    # :82-90
    ((100, None, -3, 5), (None, 1, 5)),                   # :92-100
])
def test_speaker_constraints_resolve(args, expect):
    assert oracle.speaker_constraints(*args) == expect


def test_result_is_a_lloyd_fixed_point_and_matches_sklearn_on_separated_data(oracle_mod):
    """An independent implementation (scikit-learn): (1) started from the restatement's final centroids, one more Lloyd step of sklearn
    on the L2-normalised rows (KMeansClustering.swift:65, 109 normalise first; centroids and inertia live in that space) moves nothing —
    labels and centroids are a fixed point of the algorithm KMeansClustering.swift:133-238 runs; (2) on well separated
    clusters best-of-10 finds the same partition and inertia as sklearn's own best-of-10 (the draws of the two differ by construction: the
    reference's are pinned nowhere, §2)."""
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(4)
    k, d = 6, 32
    cen = rng.standard_normal((k, d)) * 6.0
    x = cen[rng.integers(0, k, 900)] + rng.standard_normal((900, d))
    lab, c, best, inert = oracle_mod.kmeans_ninit(x, k, 100, 10, 0)
    assert c.shape == (k, d) and len(set(lab.tolist())) == k
    xn = x / np.linalg.norm(x, axis=1, keepdims=True)
    step = KMeans(n_clusters=k, init=c, n_init=1, max_iter=1, algorithm="lloyd", tol=0.0).fit(xn)
    assert np.array_equal(step.labels_, lab)
    np.testing.assert_allclose(step.cluster_centers_, c, rtol=0, atol=1e-9)
    own = KMeans(n_clusters=k, n_init=10, random_state=0, algorithm="lloyd").fit(xn)
    pairs = set(zip(lab.tolist(), own.labels_.tolist()))
    assert len(pairs) == k                                            # same partition up to the names of the clusters
    assert abs(float(np.nanmin(inert)) - own.inertia_) <= 1e-6 * own.inertia_
