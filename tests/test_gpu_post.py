"""Post-VBx centroids and cosine assignment on the device vs the CPU oracle (bit-exact: same summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,S,seed", [(500, 256, 7, 0), (1, 16, 3, 1), (3000, 128, 40, 2), (257, 33, 5, 3),
                                        (512, 100, 4, 4), (641, 64, 3, 5), (5000, 256, 12, 6), (1153, 70, 9, 7)])   # n >= 512: the tiled kernel
def test_centroids_and_assignment_match_oracle(fa, gpu_ctx, oracle_mod, n, d, S, seed):
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((n, d))
    gamma = rng.random((n, S)) ** 4
    gamma[rng.random((n, S)) < 0.3] = 0.0          # exact zeros are skipped by the reference (:655)
    gamma /= np.maximum(gamma.sum(1, keepdims=True), 1e-300)
    pi = gamma.sum(0) / n
    pi[rng.integers(0, S)] = 1e-9                  # below the 1e-7 activity threshold (:630-640)
    cr, mr = oracle_mod.weighted_centroids(emb, gamma, pi)
    cg, mg = fa.compute_centroids(emb, gamma, pi, ctx=gpu_ctx)
    np.testing.assert_array_equal(mg, mr)
    np.testing.assert_array_equal(cg, cr)
    ar = oracle_mod.assign_cosine(emb, cr)
    ag = fa.assign_embeddings(emb, cg, ctx=gpu_ctx)
    assert ag == ar.tolist()


@pytest.mark.parametrize("n", [100, 700])   # the one-wavefront kernel and the tiled one
def test_skipped_rows_never_touch_the_sums(fa, gpu_ctx, oracle_mod, n):
    """A row whose weight is not > 0 is skipped by the reference (:655) whatever its embedding holds — inf, NaN, huge values.  The device adds
    +0.0 for such a row (one add per row on the dependent chain, no select): the sums must keep the reference's bits, and a negative or NaN
    weight must count as skipped too."""
    rng = np.random.default_rng(n)
    d, S = 64, 3
    emb = rng.standard_normal((n, d))
    gamma = rng.random((n, S))
    dead = rng.choice(n, n // 5, replace=False)
    gamma[dead] = 0.0
    emb[dead[0::3]] = np.inf
    emb[dead[1::3]] = np.nan
    emb[dead[2::3]] = -1e308
    gamma[dead[0], 1] = -0.25                      # not > 0: skipped
    gamma[dead[1], 2] = np.nan                     # not > 0: skipped
    pi = np.full(S, 1.0 / S)
    cr, mr = oracle_mod.weighted_centroids(emb, gamma, pi)
    cg, mg = fa.compute_centroids(emb, gamma, pi, ctx=gpu_ctx)
    assert np.isfinite(cr).all()
    np.testing.assert_array_equal(mg, mr)
    np.testing.assert_array_equal(cg, cr)


def test_guards(fa, gpu_ctx):
    assert fa.assign_embeddings(np.zeros((0, 4)), np.zeros((2, 4)), ctx=gpu_ctx) == []
    assert fa.assign_embeddings(np.ones((3, 4)), np.zeros((0, 4)), ctx=gpu_ctx) == [0, 0, 0]
    c, m = fa.compute_centroids(np.ones((3, 4)), np.ones((3, 2)) * 0.5, np.array([1e-9, 1e-8]), ctx=gpu_ctx)
    assert c.shape == (0, 4) and m.tolist() == [-1, -1]
    # zero vectors are left unnormalised (:824-859) and score 0 against everything: first centroid wins
    assert fa.assign_embeddings(np.zeros((2, 4)), np.eye(4)[:2], ctx=gpu_ctx) == [0, 0]


def test_constrained_assignment_known_answers(fa, gpu_ctx):
    from test_oracle_assign import CONSTRAINED, HUNGARIAN
    for scores, want in HUNGARIAN:
        assert fa.HungarianAssignment.max_score_assignment(scores, ctx=gpu_ctx) == want
    assert fa.HungarianAssignment.max_score_assignment([[], []], ctx=gpu_ctx) == [-1, -1]
    for scores, chunks, want in CONSTRAINED:
        assert fa.ConstrainedClusterAssignment.assign(scores, chunks, ctx=gpu_ctx) == want
    assert fa.ConstrainedClusterAssignment.assign([], [], ctx=gpu_ctx) == []


@pytest.mark.parametrize("n,K,seed", [(3000, 7, 0), (5000, 40, 1), (600, 2, 2), (900, 130, 3)])
def test_scores_and_constrained_assignment_match_oracle(fa, gpu_ctx, oracle_mod, n, K, seed):
    rng = np.random.default_rng(seed)
    d = 64
    emb = rng.standard_normal((n, d))
    cen = rng.standard_normal((K, d))
    sr = oracle_mod.centroid_scores(emb, cen)
    sg = fa.centroid_scores(emb, cen, ctx=gpu_ctx)
    np.testing.assert_array_equal(sg, sr)
    chunks = np.repeat(np.arange((n + 2) // 3), 3)[:n]          # 3 local speakers per chunk (OfflineDiarizerTypes.swift:46-55)
    chunks = chunks[rng.permutation(n)]                          # rows of a chunk are not adjacent
    sg[rng.random(sg.shape) < 0.01] = np.nan                     # non-finite scores rank below all finite ones (:78-80)
    want = oracle_mod.constrained_assign(sg, chunks).tolist()
    got = fa.ConstrainedClusterAssignment.assign(sg, chunks, ctx=gpu_ctx)
    assert got == want
    if K >= 3:
        assert min(got) >= 0


def test_constrained_assignment_beyond_256_clusters(fa, gpu_ctx, oracle_mod):
    """More than 256 clusters (or rows in a chunk): potentials / matching live in HBM slabs instead of LDS (round 2 returned
    RUNTIME_ERROR; the reference solves any size, HungarianAssignment.swift:8-62).  Same integers -> same assignment."""
    rng = np.random.default_rng(9)
    for n, K, per in ((300, 300, 3), (640, 40, 320), (700, 257, 7)):
        scores = rng.standard_normal((n, K))
        chunks = (np.arange(n) // per).astype(np.int32)
        want = oracle_mod.constrained_assign(scores, chunks)
        got = fa.ConstrainedClusterAssignment.assign(scores, chunks, ctx=gpu_ctx)
        np.testing.assert_array_equal(np.asarray(got), want)
