"""Workspace hygiene of the linkage (N^2 * 8 B per context) and the status contract under memory pressure
(reference: FastClusterWrapper.cpp:203-243 maps std::bad_alloc to ALLOCATION_FAILURE = 4; AHCClustering.swift:52-55 degrades to
singletons on any non-zero status).  Round 2 kept every context's workspace forever and, in the batched entry, reported SUCCESS
for problems whose workspace could not be allocated."""
import numpy as np
import pytest
from conftest import speaker_mixture

pytestmark = pytest.mark.gpu
GB = 1 << 30


def ws_need(n):
    npad = (n + 255) // 256 * 256
    return npad * npad * 8


def test_cap_below_the_matrix_runs_without_a_matrix_and_a_tiny_cap_gives_singletons(fa, oracle_mod):
    """A cap below N^2 * 8 B no longer fails the call (round 4): the problem runs in the matrix-free reference-order mode — O(N d) memory like the
    reference (fastcluster_internal.hpp:1625-1800) — and returns the reference build's dendrogram (stats: reference_order == 2).  A cap below even
    that is ALLOCATION_FAILURE, and AHCClustering degrades to singletons (AHCClustering.swift:52-55)."""
    ctx = fa.Context(0)
    x = speaker_mixture(2000, 32, 5, 0.05, 1)
    sr, zr = oracle_mod.linkage_ref(x)
    ctx.set_workspace_cap(ws_need(2000) // 2)
    st, z, stats = fa.linkage(x, ctx=ctx, return_stats=True)
    assert st == sr == 0 and stats["reference_order"] == 2, (st, stats, ctx.last_error())
    np.testing.assert_array_equal(z, zr)
    assert fa.AHCClustering(ctx=ctx).cluster(x, 0.6) == oracle_mod.ahc_cluster(x, 0.6).tolist()
    ctx.set_workspace_cap(1024)
    st, z = fa.linkage(x, ctx=ctx)
    assert st == fa.ALLOCATION_FAILURE and "cap" in ctx.last_error()
    ahc = fa.AHCClustering(ctx=ctx)
    assert ahc.cluster(x, 0.6) == list(range(2000)) and ahc.last_status == fa.ALLOCATION_FAILURE   # degrade, don't crash
    ctx.set_workspace_cap(None)
    st, z, stats = fa.linkage(x, ctx=ctx, return_stats=True)
    assert st == 0 and stats["reference_order"] == 0
    np.testing.assert_array_equal(z, zr)


def test_n_beyond_any_matrix_equals_the_reference_digest(fa):
    """N = 200 000 x 4: the matrix would take 320 GB and N exceeds the block records of the filter-based rounds, where round 3 returned
    ALLOCATION_FAILURE.  Now: reference-order scans without a matrix, the dendrogram of the REFERENCE build by digest
    (tests/golden/ahc_mf_iid_200000x4.json, generated on the CPU by make_ahc_full_digest.py --d 4 --stem ...); merge pairs located on a mismatch."""
    import json
    import os
    import sys
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, gold_dir)
    from ahc_full_inputs import ahc_input, dendrogram_digest, sha256
    jp = os.path.join(gold_dir, "ahc_mf_iid_200000x4.json")
    if not os.path.exists(jp):
        pytest.skip("ahc_mf_iid_200000x4.json not committed")
    with open(jp) as f:
        gold = json.load(f)
    x = ahc_input("iid", gold["n"], gold["d"])
    assert sha256(x) == gold["input_sha256"]
    ctx = fa.Context(0)
    st, z, stats = fa.linkage(x, ctx=ctx, return_stats=True)
    assert st == 0 and stats["reference_order"] == 2 and stats["merges"] == gold["n"] - 1, (st, stats, ctx.last_error())
    dig = dendrogram_digest(z)
    if dig["dendrogram_sha256"] != gold["dendrogram_sha256"]:
        pairs = np.load(jp[:-5] + "_pairs.npz")["pairs"]
        bad = np.nonzero((z[:, :2].astype(np.int32) != pairs).any(axis=1))[0]
        raise AssertionError(f"dendrogram differs from the reference build's; first differing merge rows {bad[:5]} of {bad.size}; heights equal: "
                             f"{dig['heights_sha256'] == gold['heights_sha256']}")
    assert ctx.workspace_bytes() < (1 << 28)              # O(N d): 2 N x 4 centroids, the transpose, heap and list arrays


def test_problem_beyond_the_block_records_inside_a_batch(fa, oracle_mod):
    """A batch that contains a problem of more points than the matrix-based rounds have block records for (N > 196 608): that problem runs alone in
    the matrix-free mode — its dendrogram is the reference build's by digest —, the others stay a batch.  (Until round 5 it was marked
    ALLOCATION_FAILURE inside a batch of two or more and fa_offline_cluster_batch degraded that recording to singletons.)"""
    import json
    import os
    import sys
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, gold_dir)
    from ahc_full_inputs import ahc_input, dendrogram_digest, sha256
    jp = os.path.join(gold_dir, "ahc_mf_iid_200000x4.json")
    if not os.path.exists(jp):
        pytest.skip("ahc_mf_iid_200000x4.json not committed")
    with open(jp) as f:
        gold = json.load(f)
    big = ahc_input("iid", gold["n"], gold["d"])
    assert sha256(big) == gold["input_sha256"]
    rng = np.random.default_rng(9)
    small = [rng.standard_normal((n, gold["d"])) for n in (700, 1, 900)]
    ctx = fa.Context(0)
    st, zs, stats = fa.linkage_batch([small[0], big, small[1], small[2]], ctx=ctx, return_stats=True)
    assert st == [0, 0, 0, 0], (st, ctx.last_error())
    assert stats[1]["reference_order"] == 2 and stats[1]["merges"] == gold["n"] - 1 and stats[0]["reference_order"] == 0
    assert dendrogram_digest(zs[1])["dendrogram_sha256"] == gold["dendrogram_sha256"]
    for k, j in ((0, 0), (2, 3)):
        np.testing.assert_array_equal(zs[j], oracle_mod.linkage_ref(small[k])[1])
    ctx.close()


def test_batch_statuses_under_memory_pressure(fa, oracle_mod):
    """The combined workspace of a batch exceeds what the context may take: the batch is split until the parts fit (every dendrogram =
    the reference's); when not even one problem fits, EVERY problem reports ALLOCATION_FAILURE (round 2: SUCCESS + garbage) and the
    composed stage degrades to singleton initial clusters."""
    ctx = fa.Context(0)
    probs = [speaker_mixture(700 + 100 * k, 32, 4, 0.05, 20 + k) for k in range(6)]
    refs = [oracle_mod.linkage_ref(p)[1] for p in probs]
    total = sum(ws_need(len(p)) for p in probs)
    ctx.set_workspace_cap(total // 2)                      # the whole batch does not fit, halves do
    st, zs = fa.linkage_batch(probs, ctx=ctx)
    assert st == [0] * 6
    for z, zr in zip(zs, refs):
        np.testing.assert_array_equal(z, zr)
    ctx.set_workspace_cap(ws_need(700) // 4)               # no matrix fits: every problem ends up alone and runs without one (round 4)
    st, zs, stats = fa.linkage_batch(probs, ctx=ctx, return_stats=True)
    assert st == [0] * 6 and all(s["reference_order"] == 2 for s in stats)
    for z, zr in zip(zs, refs):
        np.testing.assert_array_equal(z, zr)
    ctx.set_workspace_cap(1024)                            # nothing at all fits
    st, zs = fa.linkage_batch(probs, ctx=ctx)
    assert st == [fa.ALLOCATION_FAILURE] * 6
    assert all(not z.any() for z in zs)                    # output untouched
    # fa_offline_cluster_batch on the same context: linkage fails -> singletons -> VBx / assignment still produce labels
    rng = np.random.default_rng(3)
    recs = []
    for k in range(3):
        n = 120 + 30 * k
        spk = rng.integers(0, 3, n)
        cen = rng.standard_normal((3, 64))
        emb = (cen[spk] + 0.05 * rng.standard_normal((n, 64))).astype(np.float32)
        rho = rng.standard_normal((3, 16))[spk] + 0.3 * rng.standard_normal((n, 16))
        recs.append((emb, rho, np.arange(n) // 3))
    phi = np.ones(16)
    ctx.set_workspace_cap(1024)
    st, out = fa.cluster_embeddings_batch(recs, phi, ctx=ctx)
    assert st == [0, 0, 0]
    for (emb, rho, ch), r in zip(recs, out):
        assert r.info["initial_clusters"] == len(emb)      # the singleton fallback of AHCClustering.swift:52-55
        ref = oracle_mod.cluster_embeddings(emb, rho, ch, phi, initial=np.arange(len(emb), dtype=np.int32))
        np.testing.assert_array_equal(np.asarray(r.assignments), ref["assignments"])


def test_limit_and_trim(fa):
    ctx = fa.Context(0)
    x = speaker_mixture(3000, 32, 5, 0.05, 2)
    assert fa.linkage(x, ctx=ctx)[0] == 0
    kept = ctx.workspace_bytes()
    assert kept >= ws_need(3000)                           # cached for the next call
    ctx.trim()
    assert ctx.workspace_bytes() == 0
    ctx.set_workspace_limit(ws_need(2000))
    assert fa.linkage(x, ctx=ctx)[0] == 0
    assert ctx.workspace_bytes() < ws_need(3000)           # larger than the limit: released when the call returned
    assert fa.linkage(x[:900], ctx=ctx)[0] == 0
    assert ctx.workspace_bytes() >= ws_need(900)           # within the limit: kept


def test_eight_pooled_contexts_at_20000_and_release_under_pressure(fa, oracle_mod):
    """8 contexts on one device, N = 20 000 each (8 x 3.2 GB of workspace), concurrently through the drop-in symbol's pool; then HBM
    is filled up to a few GB and a second context needs more than what is left: the idle workspace of the first one is released and
    the call succeeds."""
    import threading

    import torch
    pool = fa.Pool([0] * 8)
    x = speaker_mixture(20000, 64, 16, 0.03, 5)
    out = [None] * 8

    def work(i):
        with pool.acquire() as (h, _dev):
            z = np.zeros((len(x) - 1, 4))
            st = fa.lib().fa_ahc_linkage(h, x.ctypes.data, x.shape[0], x.shape[1], z.ctypes.data, z.size, 0, 0, None)
            out[i] = (st, z)
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(o is not None and o[0] == 0 for o in out)
    for o in out[1:]:
        np.testing.assert_array_equal(o[1], out[0][1])
    del pool, out
    a, b = fa.Context(0), fa.Context(0)
    n = 24000
    need = ws_need(n) + n * 64 * 8 * 6
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    filler = torch.empty(int(free - 1.6 * need), dtype=torch.uint8, device="cuda")   # room for ONE workspace of this size, not two
    y = speaker_mixture(n, 64, 16, 0.03, 6)
    st_a, z_a = fa.linkage(y, ctx=a)
    assert st_a == 0 and a.workspace_bytes() >= ws_need(n)
    st_b, z_b = fa.linkage(y, ctx=b)
    assert st_b == 0, b.last_error()
    np.testing.assert_array_equal(z_a, z_b)
    assert a.workspace_bytes() < ws_need(n)                # released by b's allocation
    del filler
    torch.cuda.empty_cache()                               # hand the filler back to the driver: later tests allocate through hipMalloc


def test_reserve_takes_the_workspace_before_the_first_request(fa, oracle_mod):
    """fa_ctx_reserve (server start-up): the linkage workspace of `recordings` problems of up to n_max x d is allocated now; the calls that follow
    find it (no growth of what the context holds) and give the reference's dendrograms; the cap applies; bad arguments are refused."""
    import ctypes as C
    ctx = fa.Context(0)
    assert ctx.workspace_bytes() == 0
    ctx.reserve(3000, 64)
    held = ctx.workspace_bytes()
    assert held >= ws_need(3000)
    x = speaker_mixture(3000, 64, 7, 0.04, 3)
    st, z = fa.linkage(x, ctx=ctx)
    sr, zr = oracle_mod.linkage_ref(x)
    assert st == sr == 0 and 0 <= ctx.workspace_bytes() - held <= x.nbytes + 4096    # only the staging of the host-pointer input was added
    np.testing.assert_array_equal(z, zr)
    ctx.reserve(3000, 64, recordings=3)                    # a batch of three shares one allocation
    held3 = ctx.workspace_bytes()
    assert held3 >= 3 * ws_need(3000)
    st, zs = fa.linkage_batch([x, x[:2900], x[:2800]], ctx=ctx)
    assert st == [0, 0, 0] and 0 <= ctx.workspace_bytes() - held3 <= 3 * x.nbytes + 4096
    np.testing.assert_array_equal(zs[0], zr)
    ctx.trim()
    # a batch large enough for two uniform batches side by side (eight recordings of >= 4 096 points): the reservation follows that dispatch — the
    # helper context's workspace is taken now as well, and the first request allocates nothing on either context (round 5)
    probs = [speaker_mixture(4300 - 20 * k, 32, 6, 0.04, 40 + k) for k in range(8)]
    ctx.reserve(4300, 32, recordings=8)
    held8 = ctx.workspace_bytes()                          # the caller's context + its helper
    assert 8 * ws_need(4300) <= held8 <= int(1.1 * 8 * ws_need(4352)) + (1 << 24)
    st, zs = fa.linkage_batch(probs, ctx=ctx)
    assert st == [0] * 8 and 0 <= ctx.workspace_bytes() - held8 <= sum(p.nbytes for p in probs) + 65536
    np.testing.assert_array_equal(zs[3], oracle_mod.linkage_ref(probs[3])[1])
    ctx.trim()
    ctx.set_workspace_cap(ws_need(3000) // 2)
    with pytest.raises(fa.FluidAudioHipError):
        ctx.reserve(3000, 64)
    assert fa.lib().fa_ctx_reserve(ctx.handle, 3000, 0, 1) == fa.INVALID_ARGUMENT
    assert fa.lib().fa_ctx_reserve(ctx.handle, 3000, 64, 0) == fa.INVALID_ARGUMENT
    assert fa.lib().fa_ctx_reserve(None, 3000, 64, 1) == fa.INVALID_ARGUMENT
    assert fa.lib().fa_ctx_reserve(ctx.handle, 1, 64, 1) == 0
