"""VBx oracle self-checks.  The reference has no numeric test of runVBx (parity unpinned), so
the restatement is checked through the invariants of the algorithm it restates
(Sources/FluidAudio/Diarizer/Offline/Clustering/VBxClustering.swift:167-664)."""
import numpy as np
from conftest import speaker_mixture


def make_problem(T=600, D=32, K=5, seed=0):
    rng = np.random.default_rng(seed)
    x = speaker_mixture(T, D, K, 0.3, seed) * 30.0  # PLDA-like scale: Fa = 0.07 needs well separated speakers
    init = (np.arange(T) % K).astype(np.int32)
    flip = rng.random(T) < 0.1
    init[flip] = rng.integers(0, K, flip.sum())
    phi = rng.uniform(0.5, 4.0, D)
    return x, init, phi


def test_vbx_invariants(oracle_mod):
    x, init, phi = make_problem()
    gamma, pi, hard, elbos = oracle_mod.vbx_refine(x, init, phi)
    np.testing.assert_allclose(gamma.sum(1), 1.0, atol=1e-12)
    assert abs(pi.sum() - 1.0) < 1e-12 and (gamma >= 0).all()
    assert len(elbos) >= 2 and np.all(np.diff(elbos) > -1e-6)  # ELBO is non-decreasing up to rounding
    assert hard.tolist() == gamma.argmax(1).tolist()
    truth = (np.arange(600) % 5)
    assert (hard == truth).mean() > 0.9  # refinement repairs the 10 % corrupted warm start


def vbx_numpy(X, init, phi, Fa=0.07, Fb=0.8, iters=20, eps=1e-4):
    """Independent vectorised restatement of runVBx (:167-664) used as a second opinion on the C oracle."""
    T, D = X.shape
    S = len(np.unique(init))
    g = np.zeros((T, S))
    g[np.arange(T), np.clip(init, 0, S - 1)] = 1
    g = np.exp(7 * g - (7 * g).max(1, keepdims=True))
    g /= g.sum(1, keepdims=True)
    g /= g.sum(1, keepdims=True)
    pi = np.full(S, 1 / S)
    phi = np.maximum(phi, 1e-12)
    rho = X * np.sqrt(phi)
    G = -0.5 * ((X ** 2).sum(1) + D * np.log(2 * np.pi))
    prev, el = -1e308, []
    for it in range(iters):
        Ns = g.sum(0)
        invL = 1 / (1 + Fa / Fb * Ns[:, None] * phi[None])
        alpha = Fa / Fb * invL * (g.T @ rho)
        phiT = ((alpha ** 2 + invL) * phi).sum(1)
        lp = Fa * (rho @ alpha.T - 0.5 * phiT[None] + G[:, None]) + np.log(np.maximum(pi, 1e-8))
        mx = lp.max(1, keepdims=True)
        e = np.exp(lp - mx)
        s = e.sum(1, keepdims=True)
        g = e / s
        ll = (mx + np.log(s)).sum()
        pi = g.sum(0)
        pi /= pi.sum()
        elbo = ll + Fb * 0.5 * (np.log(invL) - invL - alpha ** 2 + 1).sum()
        el.append(elbo)
        if it > 0 and abs(elbo - prev) < eps:
            break
        prev = elbo
    return g, pi, el


def test_c_oracle_matches_numpy_restatement(oracle_mod):
    for seed in (0, 1):
        x, init, phi = make_problem(seed=seed)
        gamma, pi, hard, elbos = oracle_mod.vbx_refine(x, init, phi)
        g2, pi2, el2 = vbx_numpy(x, init, phi)
        assert len(elbos) == len(el2)
        np.testing.assert_allclose(gamma, g2, atol=1e-12)
        np.testing.assert_allclose(pi, pi2, atol=1e-12)
        np.testing.assert_allclose(elbos, el2, rtol=1e-12)


def test_post_vbx_centroids_and_assignment(oracle_mod):
    x, init, phi = make_problem()
    gamma, pi, hard, _ = oracle_mod.vbx_refine(x, init, phi)
    cent, mp = oracle_mod.weighted_centroids(x, gamma, pi)
    assert cent.shape[0] == int((pi > 1e-7).sum())
    lab = oracle_mod.assign_cosine(x, cent)
    assert (lab == np.array([mp[h] for h in hard])).mean() > 0.95
