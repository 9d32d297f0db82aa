"""Fault injection (fa_debug_inject_fault): the degrade-don't-crash contracts of the reference, and the library's own behaviour when a host
thread or a device allocation is not to be had.  Every case must return what the undisturbed path (or the reference's documented degrade)
returns — bit for bit."""
import ctypes as C

import numpy as np
import pytest
from test_gpu_pipeline import synth_session

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _disarm(fa):
    yield
    for site in range(5):
        fa.lib().fa_debug_inject_fault(site, 0)


def test_vbx_failure_degrades_to_the_initial_clusters(fa, gpu_ctx, oracle_mod):
    """VBxClustering.refine's catch block (VBxClustering.swift:136-141): gamma := one-hot initial labels, pi := 1/S, ELBOs := [];
    OfflineDiarizerManager.cluster goes on to centroids and assignment with those."""
    emb, rho, chunks, phi, _ = synth_session(120, 5, 2)
    fa.lib().fa_debug_inject_fault(fa._lib.FAULT_VBX, 1)
    one = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=gpu_ctx, intermediates=True)
    ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi, vbx_fails=True)
    assert one.info["vbx_degraded"] == 1 and one.info["vbx_iterations"] == 0 and one.info["ahc_degraded"] == 0
    assert one.info["elbos"].size == 0
    assert np.array_equal(one.info["vbx_hard"], ref["hard"]) and np.array_equal(one.initial_clusters, ref["initial"])
    assert np.asarray(one.assignments).tolist() == ref["assignments"].tolist()
    np.testing.assert_allclose(one.centroids, ref["centroids"], rtol=0, atol=1e-12)
    # a longer recording: the one-hot posteriors (exact zeros in every row) through the tiled centroid sums, whose zero-weight rows are ADDED as +-0 instead of
    # skipped when every training row is finite (round 6) — the reference's bits (it skips them)
    emb2, rho2, chunks2, phi2, _ = synth_session(260, 4, 0)
    fa.lib().fa_debug_inject_fault(fa._lib.FAULT_VBX, 1)
    two = fa.cluster_embeddings(emb2, rho2, chunks2, phi2, ctx=gpu_ctx, intermediates=True)
    ref2 = oracle_mod.cluster_embeddings(emb2, rho2, chunks2, phi2, vbx_fails=True)
    assert two.info["vbx_degraded"] == 1 and len(emb2) >= 512
    np.testing.assert_array_equal(two.centroids, ref2["centroids"])
    assert np.asarray(two.assignments).tolist() == ref2["assignments"].tolist()
    # the next call is undisturbed
    again = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=gpu_ctx)
    assert again.info["vbx_degraded"] == 0 and again.info["vbx_iterations"] > 0
    assert again.assignments == oracle_mod.cluster_embeddings(emb, rho, chunks, phi)["assignments"].tolist()


def test_vbx_refine_entry_degrades(fa, gpu_ctx, oracle_mod):
    """fa_vbx_refine is VBxClustering.refine: on an internal failure it returns the degrade, not an error."""
    emb, rho, chunks, phi, _ = synth_session(50, 3, 4)
    initial = oracle_mod.ahc_cluster(emb.astype(np.float64), 0.6)
    fa.lib().fa_debug_inject_fault(fa._lib.FAULT_VBX, 1)
    out = fa.VBxClustering(phi).refine(rho, initial)
    g, pi, hard, elbos = oracle_mod.vbx_refine_degraded(initial)
    assert np.array_equal(np.asarray(out.gamma), g) and np.array_equal(np.asarray(out.pi), pi)
    assert np.asarray(out.hard_clusters).reshape(-1).tolist() == hard.tolist() and len(out.elbos) == 0
    assert "degraded" in gpu_ctx.last_error()


def test_linkage_failure_degrades_to_singletons(fa, gpu_ctx, oracle_mod):
    """AHCClustering.swift:52-55: a non-zero status of the wrapper -> labels 0 ..< N; VBx then starts from N speakers."""
    emb, rho, chunks, phi, _ = synth_session(20, 3, 5)
    fa.lib().fa_debug_inject_fault(fa._lib.FAULT_AHC, 1)
    one = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=gpu_ctx, intermediates=True)
    ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi, ahc_fails=True)
    assert one.info["ahc_degraded"] == 1 and one.info["initial_clusters"] == len(emb)
    assert np.array_equal(one.initial_clusters, np.arange(len(emb)))
    assert np.asarray(one.assignments).tolist() == ref["assignments"].tolist()


def test_no_host_thread_to_be_had(fa, gpu_ctx, oracle_mod):
    """std::thread construction failing (std::system_error) must not unwind a vector of joinable threads across the C ABI: the share of the
    missing thread runs on the calling thread — batch clustering, the sharded CTC entry and the many-recordings linkage give their usual bits."""
    recs = []
    for r in range(5):
        emb, rho, chunks, phi, _ = synth_session(40 + 10 * r, 3 + r % 2, 20 + r)
        recs.append((emb, rho, chunks))
    base_st, base = fa.cluster_embeddings_batch(recs, phi, ctx=gpu_ctx)
    fa.lib().fa_debug_inject_fault(fa._lib.FAULT_THREAD_START, 1000)
    st, res = fa.cluster_embeddings_batch(recs, phi, ctx=gpu_ctx)
    assert list(st) == list(base_st) == [0] * 5
    for a, b in zip(res, base):
        assert list(a.assignments) == list(b.assignments) and np.array_equal(a.centroids, b.centroids)
    # sharded entries over a pool that lists device 0 twice
    pool = fa.Pool([0, 0])
    try:
        rng = np.random.default_rng(0)
        lg = rng.standard_normal((6, 200, 129)).astype(np.float32)
        lg[:, :, 128] += 2
        ids, lens = pool.ctc_greedy_batch(lg, 128)
        for b in range(6):
            assert ids[b, :lens[b]].tolist() == oracle_mod.ctc_greedy(lg[b], 128).tolist()
        xs = [oracle_mod.ahc_normalize(rng.standard_normal((150 + 20 * k, 32))) for k in range(3)]
        sts, zs = pool.linkage_many(xs)
        for x, s, z in zip(xs, sts, zs):
            sr, zr = oracle_mod.linkage_ref(x)
            assert s == sr == 0 and np.array_equal(z, zr)
    finally:
        fa.lib().fa_debug_inject_fault(fa._lib.FAULT_THREAD_START, 0)
        pool.close()


def test_allocation_failure_releases_every_idle_cache_on_the_device(fa, gpu_ctx, oracle_mod):
    """A failing hipMalloc (of a cached-buffer request or of a linkage workspace) first releases what the library holds idle on that device:
    the buffer caches of ALL contexts (the caller's, its workers', other contexts') and the idle linkage workspaces — then retries."""
    L = fa.lib()
    other = fa.Context(0)
    emb, rho, chunks, phi, _ = synth_session(60, 3, 9)
    base = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=other)          # leaves buffers in `other`'s cache and a linkage workspace
    assert L.fa_ctx_workspace_bytes(other.handle) > 0
    x = oracle_mod.ahc_normalize(np.random.default_rng(1).standard_normal((300, 64)))
    for site in (fa._lib.FAULT_DEVBUF_MALLOC, fa._lib.FAULT_WS_MALLOC):
        fa.cluster_embeddings(emb, rho, chunks, phi, ctx=other)
        before = L.fa_ctx_workspace_bytes(other.handle)
        assert before > 0
        L.fa_debug_inject_fault(site, 1)
        if site == fa._lib.FAULT_WS_MALLOC:
            gpu_ctx.trim()                                                   # so that the linkage below needs a fresh workspace
            st, z = fa.linkage(x, ctx=gpu_ctx)
            sr, zr = oracle_mod.linkage_ref(x)
            assert st == sr == 0 and np.array_equal(z, zr)
        else:
            gpu_ctx.trim()                                                   # empty cache: the request reaches hipMalloc
            res = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=gpu_ctx)
            assert res.assignments == base.assignments
        after = L.fa_ctx_workspace_bytes(other.handle)
        assert after < before, (site, before, after)                        # `other` gave its idle memory up
    # a workspace limit of 0 keeps nothing cached — the buffer cache included
    lim = fa.Context(0)
    L.fa_ctx_set_workspace_limit(lim.handle, 0)
    fa.cluster_embeddings(emb, rho, chunks, phi, ctx=lim)
    assert L.fa_ctx_workspace_bytes(lim.handle) <= 1 << 20                  # at most the small scratch buffer
    lim.close()
    other.close()
