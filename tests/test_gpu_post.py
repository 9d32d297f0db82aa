"""Post-VBx centroids and cosine assignment on the device vs the CPU oracle (bit-exact: same summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,S,seed", [(500, 256, 7, 0), (1, 16, 3, 1), (3000, 128, 40, 2), (257, 33, 5, 3)])
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


def test_guards(fa, gpu_ctx):
    assert fa.assign_embeddings(np.zeros((0, 4)), np.zeros((2, 4)), ctx=gpu_ctx) == []
    assert fa.assign_embeddings(np.ones((3, 4)), np.zeros((0, 4)), ctx=gpu_ctx) == [0, 0, 0]
    c, m = fa.compute_centroids(np.ones((3, 4)), np.ones((3, 2)) * 0.5, np.array([1e-9, 1e-8]), ctx=gpu_ctx)
    assert c.shape == (0, 4) and m.tolist() == [-1, -1]
    # zero vectors are left unnormalised (:824-859) and score 0 against everything: first centroid wins
    assert fa.assign_embeddings(np.zeros((2, 4)), np.eye(4)[:2], ctx=gpu_ctx) == [0, 0]
