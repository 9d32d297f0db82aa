"""GPU parity: centroid-linkage AHC (HIP, through the C ABI incl. the drop-in symbol) vs the reference's own C++
build (oracle/_ref) and committed golden dendrograms — bit-exact dendrograms on tie-free data, identical partitions
(and identical height multisets) where exact ties make the reference's merge order a heap artefact."""
import os

import numpy as np
import pytest
from conftest import same_partition, speaker_mixture

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ahc_golden.npz")


@pytest.mark.parametrize("mode", [0, 1])
def test_committed_golden_dendrograms(fa, gpu_ctx, mode):
    g = np.load(GOLD)
    for x, z, lab, thr in ((g["xm"], g["zm"], g["labels_m"], 0.6), (g["xi"], g["zi"], g["labels_i"], 1.2)):
        st, got, stats = fa.linkage(x, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st == 0
        np.testing.assert_array_equal(got, z)  # bit-exact: ids, heights, sizes, merge order
        assert stats["merges"] == x.shape[0] - 1
        np.testing.assert_array_equal(fa.cut(got, x.shape[0], thr), lab)


def test_drop_in_symbol_status_contract(fa, gpu_ctx):
    f = fa.lib().fastcluster_compute_centroid_linkage
    x = np.random.default_rng(0).standard_normal((5, 3))
    z = np.zeros(16)
    assert f(x.ctypes.data, 5, 3, z.ctypes.data, 16) == 0 and z[3] == 2.0
    assert f(x.ctypes.data, 5, 3, z.ctypes.data, 15) == 3
    assert f(None, 5, 3, z.ctypes.data, 16) == 1
    assert f(x.ctypes.data, 0, 3, z.ctypes.data, 16) == 0
    assert f(x.ctypes.data, 5, 0, z.ctypes.data, 16) == 1
    assert f(x.ctypes.data, 1, 3, z.ctypes.data, 0) == 0
    x[2, 1] = np.nan
    assert f(x.ctypes.data, 5, 3, z.ctypes.data, 16) == 5  # NaN distance -> RUNTIME_ERROR (FastClusterWrapper.cpp:236-237)
    st, z2 = fa.fastcluster_compute_centroid_linkage(np.eye(3))
    assert st == 0 and z2.shape == (2, 4)


@pytest.mark.parametrize("n,d,kind", [(2, 4, "iid"), (3, 1, "iid"), (257, 7, "iid"), (1000, 256, "iid"), (1500, 64, "mix"),
                                      (2000, 256, "mix"), (777, 300, "iid"), (600, 16, "iid"), (900, 48, "mix"),
                                      (400, 257, "iid"), (300, 255, "iid"), (520, 129, "mix")])   # odd d: centroid rows on 8-byte boundaries, a last element without a partner (round 6)   # d % 16 == 0: the direct-to-LDS Gram kernel (one and three k-chunks)
def test_bit_exact_vs_reference_build(fa, gpu_ctx, oracle_mod, n, d, kind):
    if kind == "iid":
        x = oracle_mod.ahc_normalize(np.random.default_rng(n).standard_normal((n, d)))
    else:
        x = speaker_mixture(n, d, 16, 0.03, n)
    sr, zr = oracle_mod.linkage_ref(x)
    assert sr == 0
    for mode in (fa.AHC_MODE_AUTO, fa.AHC_MODE_EXACT):
        st, z, stats = fa.linkage(x, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st == 0
        np.testing.assert_array_equal(z, zr)
    # unnormalised, shifted data (the ABI does not require unit rows)
    y = x * 3.0 + 0.5
    _, zr = oracle_mod.linkage_ref(y)
    st, z = fa.linkage(y, ctx=gpu_ctx)
    assert st == 0
    np.testing.assert_array_equal(z, zr)


def test_swift_known_answers_on_device(fa, gpu_ctx):
    ahc = fa.AHCClustering(ctx=gpu_ctx)                     # AHCClusteringTests.swift:12-145
    assert ahc.cluster([], 0.7) == []
    assert ahc.cluster([[1.0, 0.0, 0.0]], 0.7) == [0]
    assert len(set(ahc.cluster([[1.0, 2.0, 3.0]] * 5, 0.7))) == 1
    g = [[1, 0, 0], [.9, .1, 0], [.95, .05, 0], [0, 1, 0], [0, .9, .1], [0, .95, .05]]
    r = ahc.cluster(g, 0.8)
    assert len(set(r[:3])) == 1 and len(set(r[3:])) == 1 and r[0] != r[3]
    e4 = [[1, 0, 0], [.9, .1, 0], [0, 1, 0], [0, .9, .1]]
    assert len(set(ahc.cluster(e4, 0.5))) == 2 and len(set(ahc.cluster(e4, 1.5))) == 1
    eye = np.eye(3).tolist()
    assert sorted(set(ahc.cluster(eye, 0.5))) == [0, 1, 2]
    assert len(set(ahc.cluster(eye, 2.0))) == 1 and len(set(ahc.cluster(eye, 0.0))) == 3
    assert ahc.cluster([[], [], []], 0.7) == [0, 0, 0]
    bad = [[1.0, float("nan")], [0.0, 1.0], [1.0, 1.0]]
    assert ahc.cluster(bad, 0.7) == [0, 1, 2] and ahc.last_status == 5  # degrade to singletons (:52-55)


def test_exact_ties_same_heights_and_partitions(fa, gpu_ctx, oracle_mod):
    """Duplicated / symmetric inputs: the reference's merge ORDER among bit-identical distances is an artefact of
    its heap; heights (as a multiset) and the partition at every threshold must agree."""
    rng = np.random.default_rng(3)
    base = oracle_mod.ahc_normalize(rng.standard_normal((40, 16)))
    cases = [np.repeat(base, 3, axis=0),                                  # every point three times
             np.array([[1, 0, 0], [.9, .1, 0], [.95, .05, 0], [0, 1, 0], [0, .9, .1], [0, .95, .05]], float),
             np.vstack([np.eye(8), np.eye(8)])]
    for x in cases:
        xn = oracle_mod.ahc_normalize(x)
        _, zr = oracle_mod.linkage_ref(xn)
        for mode in (fa.AHC_MODE_AUTO, fa.AHC_MODE_REFERENCE_ORDER, fa.AHC_MODE_EXACT):
            st, z, stats = fa.linkage(xn, mode=mode, ctx=gpu_ctx, return_stats=True)
            assert st == 0
            if mode != fa.AHC_MODE_EXACT:      # round 3: the reference's own order among the ties (csrc/ahc_reforder.h)
                np.testing.assert_array_equal(z, zr)
                assert stats["reference_order"] == 1 or mode == fa.AHC_MODE_AUTO   # AUTO takes that route only when it meets a tie at the minimum
            np.testing.assert_allclose(np.sort(z[:, 2]), np.sort(zr[:, 2]), rtol=0, atol=1e-15)
            for thr in (0.0, 1e-9, 0.3, 0.6, 1.0, 1.3, 1.5, 2.0):
                assert same_partition(fa.cut(z, len(xn), thr), oracle_mod.ahc_cut(zr, len(xn), thr)), (thr, mode)
        # the Lance-Williams filter must have detected the ambiguity and handed over to exact rows
    assert stats is not None


def test_cluster_labels_bit_exact_config3_style(fa, gpu_ctx, oracle_mod):
    """Config-3 distributions at oracle-friendly size: labels equal the reference pipeline at several thresholds."""
    n = 3000
    for x in (np.random.default_rng(0).standard_normal((n, 256)), speaker_mixture(n, 256, 64, 0.02, 0) * 1.7):
        for thr in (0.6, 1.0, 1.05, 1.2):
            got = fa.AHCClustering(ctx=gpu_ctx).cluster(x, thr)
            ref = oracle_mod.ahc_cluster(x, thr)
            np.testing.assert_array_equal(np.asarray(got, np.int32), ref)
    assert len(set(fa.AHCClustering(ctx=gpu_ctx).cluster(speaker_mixture(n, 256, 64, 0.02, 0), 0.6))) == 64


def test_device_pointer_entry_and_reuse(fa, gpu_ctx, oracle_mod):
    import ctypes as C

    import torch
    x = speaker_mixture(900, 32, 9, 0.05, 4)
    _, zr = oracle_mod.linkage_ref(x)
    d_x = torch.from_numpy(x).cuda()
    d_z = torch.zeros((899, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):  # second call reuses the cached workspace
        st = fa.lib().fa_ahc_linkage(gpu_ctx.handle, C.c_void_p(d_x.data_ptr()), 900, 32, C.c_void_p(d_z.data_ptr()), 899 * 4,
                                     fa.AHC_MODE_AUTO, 1, None)
        assert st == 0
        np.testing.assert_array_equal(d_z.cpu().numpy(), zr)


def test_full_size_config3_known_prefix_and_structure(fa, gpu_ctx):
    """BASELINE config 3 size (50 000 x 256), checked through size-independent properties: half of the dendrogram has a
    closed-form answer (25 000 planted pairs whose members differ in ONE coordinate, so the centroid-linkage height is
    exactly |delta| and the greedy order is the order of the deltas), the other half must be a structurally valid
    dendrogram, and a second run must reproduce the first bit for bit."""
    n, d = 50000, 256
    rng = np.random.default_rng(11)
    base = rng.standard_normal((n // 2, d))
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    pos = rng.permutation(n).reshape(-1, 2)                       # the two rows of pair i
    x = np.empty((n, d))
    x[pos[:, 0]] = base
    x[pos[:, 1]] = base
    coord = rng.integers(0, d, n // 2)
    delta = np.linspace(1e-4, 0.2, n // 2)[rng.permutation(n // 2)]
    x[pos[:, 1], coord] += delta
    height = np.abs(x[pos[:, 1], coord] - x[pos[:, 0], coord])    # sqrt of a single square is exact
    assert len(np.unique(height)) == n // 2
    st, z, stats = fa.linkage(x, ctx=gpu_ctx, return_stats=True)
    assert st == 0
    order = np.argsort(height)
    want = np.stack([pos[order].min(1), pos[order].max(1), height[order], np.full(n // 2, 2.0)], axis=1)
    np.testing.assert_array_equal(z[: n // 2], want)
    # structure of the whole dendrogram (FastClusterWrapper.cpp:169-192): every node is merged exactly once, children precede
    # their parent, sizes add up
    a, b = z[:, 0].astype(np.int64), z[:, 1].astype(np.int64)
    assert (a < b).all() and (b < n + np.arange(n - 1)).all() and (a >= 0).all()
    assert np.array_equal(np.sort(np.concatenate([a, b])), np.arange(2 * n - 2))
    size = np.concatenate([np.ones(n), z[:, 3]])
    np.testing.assert_array_equal(z[:, 3], size[a] + size[b])
    assert z[-1, 3] == n and np.isfinite(z[:, 2]).all() and (z[:, 2] >= 0).all()
    assert (z[n // 2:, 2] > 0.5).all()                            # pair centroids are ~sqrt(2) apart; centroid linkage may invert but not collapse
    st2, z2 = fa.linkage(x, ctx=gpu_ctx)
    assert st2 == 0 and np.array_equal(z, z2)
    labels = fa.cut(z, n, 0.25)
    assert len(set(labels.tolist())) == n // 2 and (labels[pos[:, 0]] == labels[pos[:, 1]]).all()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] at FULL size against the reference itself: tests/golden/ahc_full_<dist>_<n>.json hold SHA-256 digests
# of what the reference's own C++ (oracle/_ref, built from /root/reference) returned for the seeded inputs of
# tests/golden/ahc_full_inputs.py (generated by tests/golden/make_ahc_full_digest.py, ~15 CPU-minutes per 50 k run).
import glob  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
FULL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ahc_full_*_*.json")))


def _first_mismatch(z, stem):
    """Row of the first differing merge pair, from the committed pair list (diagnostics for a failing digest)."""
    ref = np.load(stem + "_pairs.npz")["pairs"]
    bad = np.nonzero((z[:, :2].astype(np.int32) != ref).any(axis=1))[0]
    return None if bad.size == 0 else (int(bad[0]), z[bad[0]].tolist(), ref[bad[0]].tolist())


@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "exact"])
@pytest.mark.parametrize("path", FULL, ids=[os.path.basename(p)[9:-5] for p in FULL])
def test_full_size_dendrogram_digest_vs_reference(fa, gpu_ctx, path, mode):
    """Bit-exact at the configured size: dendrogram bytes (merge order, ids, heights, sizes) and the label vectors after
    AHCClustering's cut at thr 0.6 / 1.0 / 1.05 / 1.2 hash to what the reference build produced — in AUTO mode (Gram
    start-up, Lance-Williams filter + exact certification windows: the path only large N exercises) and in EXACT mode."""
    from ahc_full_inputs import ahc_input, dendrogram_digest, sha256
    want = json.load(open(path))
    n, d = want["n"], want["d"]
    x = ahc_input(want["dist"], n, d)
    assert sha256(x) == want["input_sha256"], "the seeded input did not regenerate bit-for-bit (numpy version?)"
    st, z, stats = fa.linkage(x, mode=mode, ctx=gpu_ctx, return_stats=True)
    assert st == 0
    got = dendrogram_digest(z)
    if got["dendrogram_sha256"] != want["dendrogram_sha256"]:
        pytest.fail(f"dendrogram differs from the reference: pairs equal {got['pairs_sha256'] == want['pairs_sha256']}, heights equal "
                    f"{got['heights_sha256'] == want['heights_sha256']}, first differing merge {_first_mismatch(z, path[:-5])}, stats {stats}")
    for thr, c in want["cuts"].items():
        lab = fa.cut(z, n, float(thr))
        assert int(lab.max()) + 1 == c["clusters"]
        assert sha256(lab.astype(np.int32)) == c["labels_sha256"], f"labels at thr {thr}"
    print(f"{os.path.basename(path)} mode {mode}: bit-exact vs the reference; {stats}")


def test_full_size_digests_are_committed():
    names = {os.path.basename(p) for p in FULL}
    assert {"ahc_full_iid_50000.json", "ahc_full_mix_50000.json"} <= names, "the 50 000 x 256 reference digests are missing"


def test_batched_problems_equal_single_problem_runs(fa, gpu_ctx, oracle_mod):
    """fa_ahc_linkage_batch: K independent recordings advanced by the same round launches give, per recording, the dendrogram
    of the reference build bit for bit (ragged sizes, both distributions, a 1-row, a 2-row and an empty problem, a NaN problem
    that must fail alone)."""
    rng = np.random.default_rng(3)
    probs = [speaker_mixture(700, 64, 9, 0.04, 1), oracle_mod.ahc_normalize(rng.standard_normal((1300, 64))), speaker_mixture(257, 64, 5, 0.05, 2),
             np.ones((1, 64)), np.eye(2, 64), np.zeros((0, 64)), speaker_mixture(2100, 64, 20, 0.03, 4), speaker_mixture(300, 64, 4, 0.05, 6)]
    bad = speaker_mixture(300, 64, 4, 0.05, 5).copy()
    bad[17, 3] = np.nan
    probs.append(bad)
    for mode in (0, 1):
        st, zs, stats = fa.linkage_batch(probs, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st[:-1] == [0] * (len(probs) - 1) and st[-1] == 5           # NaN -> RUNTIME_ERROR for that problem only
        for x, z in zip(probs[:-1], zs[:-1]):
            if x.shape[0] >= 2:
                sr, zr = oracle_mod.linkage_ref(x)
                assert sr == 0
                np.testing.assert_array_equal(z, zr)
        assert stats[0]["merges"] == 699 and stats[6]["merges"] == 2099
    # and equal to the single-problem entry (same kernels, same order)
    st1, z1 = fa.linkage(probs[6], ctx=gpu_ctx)
    assert st1 == 0 and np.array_equal(z1, zs[6])


def test_row_minima_slabs_equal_restatement_and_linkage_first_merge(fa, gpu_ctx, oracle_mod):
    """fa_ahc_row_minima (the shardable start-up table): slabs concatenate to the full table, the table equals the CPU
    restatement (same sequential sums -> same bits), and its global minimum is the first merge of the reference build."""
    import ctypes as C
    from fluidaudio_amd.sharding import row_minima_numpy
    x = speaker_mixture(1100, 96, 11, 0.05, 8)
    x[300] = x[77]                                         # exact tie
    n, d = x.shape
    def slab(lo, hi):
        m, a = np.zeros(hi - lo), np.zeros(hi - lo, np.int32)
        gpu_ctx.check(fa.lib().fa_ahc_row_minima(gpu_ctx.handle, x.ctypes.data, n, d, lo, hi, m.ctypes.data, a.ctypes.data, 0), "fa_ahc_row_minima")
        return m, a
    full_m, full_a = slab(0, n)
    parts = [slab(lo, hi) for lo, hi in ((0, 137), (137, 138), (138, 700), (700, n))]
    np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), full_m)
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), full_a)
    rm, ra = row_minima_numpy(x, 0, n)
    np.testing.assert_array_equal(full_a, ra)
    np.testing.assert_array_equal(full_m, rm)
    assert full_a[300] == 77 and full_a[77] == 300 and full_m[300] == 0.0
    y = speaker_mixture(900, 64, 7, 0.05, 9)               # tie-free: the closest pair is the reference's first merge
    m, a = np.zeros(900), np.zeros(900, np.int32)
    gpu_ctx.check(fa.lib().fa_ahc_row_minima(gpu_ctx.handle, y.ctypes.data, 900, 64, 0, 900, m.ctypes.data, a.ctypes.data, 0), "fa_ahc_row_minima")
    st, z = oracle_mod.linkage_ref(y)
    i = int(np.argmin(m))
    assert st == 0 and {int(z[0, 0]), int(z[0, 1])} == {i, int(a[i])} and z[0, 2] == np.sqrt(m[i])


def test_batch_of_more_problems_than_fit_the_kernel_arguments(fa, gpu_ctx, oracle_mod):
    """Up to 16 running problems travel in the kernel arguments (ahc_round_args); more go through the table / block map in HBM and
    drop to the argument form once enough of them have finished.  20 small problems of different sizes: every dendrogram = the
    reference build's."""
    probs = [speaker_mixture(150 + 37 * k, 32, 3 + k % 4, 0.05, 40 + k) for k in range(20)]
    st, zs = fa.linkage_batch(probs, ctx=gpu_ctx)
    assert st == [0] * 20
    for x, z in zip(probs, zs):
        sr, zr = oracle_mod.linkage_ref(x)
        assert sr == 0
        np.testing.assert_array_equal(z, zr)


def test_massive_exact_ties_reference_order_vs_exact_mode(fa, gpu_ctx):
    """30 % / 90 % of the rows are exact copies of other rows: the Lance-Williams filter cannot certify anything, the window
    overflows and the problem is recomputed in the reference's selection order (reference_order) — same multiset of heights and the
    same partition as a run in exact mode (its own tie order) from the start."""
    for n, dup in ((3000, 0.3), (4000, 0.9)):
        x = speaker_mixture(n, 64, 12, 0.03, 7).copy()
        rng = np.random.default_rng(1)
        x[rng.integers(0, n, int(n * dup))] = x[rng.integers(0, n, int(n * dup))]
        st0, z0, s0 = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)
        st1, z1, s1 = fa.linkage(x, mode=fa.AHC_MODE_EXACT, ctx=gpu_ctx, return_stats=True)
        assert st0 == st1 == 0 and s0["exact_fallback"] == 1 and s0["reference_order"] == 1 and s0["rounds"] <= 8 * n
        np.testing.assert_array_equal(np.sort(z0[:, 2]), np.sort(z1[:, 2]))
        assert same_partition(fa.cut(z0, n, 0.6), fa.cut(z1, n, 0.6))


def test_more_than_65536_points_takes_the_many_record_path(fa, gpu_ctx):
    """N > 65 536: more than four block records per lane in the round's first reduction (the generic path of ahc_round_body; every
    other test stays below it).  66 000 x 4: the filter-based rounds (AUTO) against the reference-order scans (a different set of
    kernels that evaluates every distance exactly, checked against the reference build elsewhere) — row for row.  35 GB of workspace."""
    import torch
    n, d = 66_000, 4
    x = np.random.default_rng(66).standard_normal((n, d))
    st, z, stats = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)
    assert st == 0 and stats["merges"] == n - 1 and stats["reference_order"] == 0, stats
    st2, z2, stats2 = fa.linkage(x, mode=fa.AHC_MODE_REFERENCE_ORDER, ctx=gpu_ctx, return_stats=True)
    assert st2 == 0 and stats2["reference_order"] == 1
    bad = np.nonzero((z != z2).any(axis=1))[0]
    assert bad.size == 0, f"first differing row {bad[:3]}: {z[bad[:1]]} vs {z2[bad[:1]]}; stats {stats}"
    gpu_ctx.trim()
    torch.cuda.empty_cache()


def test_batch_of_large_problems_three_ways(fa, gpu_ctx, switch):
    """fa_ahc_linkage_batch with 2 .. 4 problems of >= 16 384 points: by default ONE launch per round advances all of them (uniform layout,
    ahc_round_uni; round 4); FA_AHC_IN_FLIGHT=1 runs each merge chain on its own helper context concurrently (round 3), FA_AHC_NO_UNIFORM=1 the
    round-2 batched chain.  Every dendrogram of every form equals the single-problem call bit for bit; helper workspaces count towards
    fa_ctx_workspace_bytes and go with fa_ctx_trim."""
    import torch
    rng = np.random.default_rng(77)
    probs = [rng.standard_normal((n, 16)) for n in (17000, 16500, 18000)]
    singles = [fa.linkage(x, ctx=gpu_ctx)[1] for x in probs]
    gpu_ctx.trim()
    st, zs, stats = fa.linkage_batch(probs, ctx=gpu_ctx, return_stats=True)
    assert list(st) == [0, 0, 0]
    for z, zr, s in zip(zs, singles, stats):
        np.testing.assert_array_equal(z, zr)
        assert s["merges"] == len(zr)
    one = 18176 * 18176 * 8
    assert gpu_ctx.workspace_bytes() > 3 * one             # three workspaces of the largest problem's layout, one allocation
    gpu_ctx.trim()
    switch("FA_AHC_IN_FLIGHT", "1")
    st1, zs1 = fa.linkage_batch(probs, ctx=gpu_ctx)
    assert list(st1) == [0, 0, 0]
    for z, zr in zip(zs1, singles):
        np.testing.assert_array_equal(z, zr)
    assert gpu_ctx.workspace_bytes() > 2.5 * 17000 * 17000 * 8   # the context's and two helpers'
    switch("FA_AHC_IN_FLIGHT", None)
    switch("FA_AHC_NO_UNIFORM", "1")
    st2, zs2 = fa.linkage_batch(probs, ctx=gpu_ctx)
    for z, zr in zip(zs2, singles):
        np.testing.assert_array_equal(z, zr)
    gpu_ctx.trim()
    assert gpu_ctx.workspace_bytes() < (1 << 26)
    torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["", "6", "8"])
def test_uniform_batch_equals_reference_build(fa, gpu_ctx, oracle_mod, switch, waves):
    """The uniform-layout batch (every problem in the layout of the largest, one launch per round, problem = workgroup id y): ragged sizes
    within a factor of two, both distributions, a NaN problem that fails alone, a problem with exact ties at the minimum that is recomputed in
    reference order — per problem the reference build's dendrogram bit for bit, at each of the kernel's three register budgets."""
    if waves:
        switch("FA_AHC_UNI_WAVES", waves)
    rng = np.random.default_rng(11)
    tied = speaker_mixture(1400, 64, 6, 0.05, 21).copy()
    tied[700:1400] = tied[0:700]                            # every row twice: exact ties at every minimum
    probs = [speaker_mixture(2100, 64, 20, 0.03, 4), oracle_mod.ahc_normalize(rng.standard_normal((1300, 64))), speaker_mixture(2047, 64, 9, 0.04, 1),
             speaker_mixture(1537, 64, 5, 0.05, 2), tied]
    bad = speaker_mixture(1500, 64, 4, 0.05, 5).copy()
    bad[17, 3] = np.nan
    probs.append(bad)
    for mode in (0, 1):
        st, zs, stats = fa.linkage_batch(probs, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st[:-1] == [0] * (len(probs) - 1) and st[-1] == 5, st
        for k, (x, z) in enumerate(zip(probs[:-1], zs[:-1])):
            sr, zr = oracle_mod.linkage_ref(x)
            assert sr == 0
            if k == 4 and mode == 1:                        # exact mode keeps its own order among exact ties: heights and partition
                np.testing.assert_array_equal(np.sort(z[:, 2]), np.sort(zr[:, 2]))
                continue
            np.testing.assert_array_equal(z, zr)
        assert stats[0]["merges"] == 2099 and stats[3]["merges"] == 1536
        if mode == 0:
            assert stats[4]["reference_order"] == 1 and stats[0]["reference_order"] == 0
    # two equal problems next to a different one: no cross-talk between the workspaces
    st, zs = fa.linkage_batch([probs[0], probs[2], probs[0]], ctx=gpu_ctx)
    assert st == [0, 0, 0]
    np.testing.assert_array_equal(zs[0], zs[2])
    st1, z1 = fa.linkage(probs[2], ctx=gpu_ctx)
    np.testing.assert_array_equal(zs[1], z1)


def test_uniform_batches_side_by_side_equal_single_calls(fa, gpu_ctx, switch):
    """Six or more large recordings: two uniform batches on two streams (the caller's context and a helper context) fill each other's latency
    (ahc_batch_uniform_groups; FA_AHC_UNI_GROUPS picks 1 .. 4 groups).  Every dendrogram equals the single call at every group count, the statistics
    are those of the problem's own group, and a cap that leaves no room for the groups' workspaces still ends in the right dendrograms."""
    import torch
    rng = np.random.default_rng(78)
    probs = [rng.standard_normal((n, 16)) for n in (17000, 16500, 18000, 16400, 17500, 16900)]
    singles = [fa.linkage(x, ctx=gpu_ctx)[1] for x in probs]
    gpu_ctx.trim()
    for groups in (None, "1", "2", "3"):
        if groups is None:
            switch("FA_AHC_UNI_GROUPS", None)
        else:
            switch("FA_AHC_UNI_GROUPS", groups)
        st, zs, stats = fa.linkage_batch(probs, ctx=gpu_ctx, return_stats=True)
        assert list(st) == [0] * 6, (groups, st)
        for z, zr, s in zip(zs, singles, stats):
            np.testing.assert_array_equal(z, zr)
            assert s["merges"] == len(zr) and s["reference_order"] == 0
    # a recording with exact ties inside the SECOND group (it runs on the helper context): recomputed there in reference order, the others untouched
    tied = probs[4].copy()
    tied[8000:16000] = tied[0:8000]
    switch("FA_AHC_UNI_GROUPS", "2")
    st, zs, stats = fa.linkage_batch(probs[:4] + [tied] + probs[5:], ctx=gpu_ctx, return_stats=True)
    assert list(st) == [0] * 6 and stats[4]["reference_order"] == 1 and stats[3]["reference_order"] == 0
    st1, z1 = fa.linkage(tied, ctx=gpu_ctx)
    np.testing.assert_array_equal(zs[4], z1)
    np.testing.assert_array_equal(zs[5], singles[5])
    gpu_ctx.trim()
    ctx = fa.Context(0)
    ctx.set_workspace_cap(int(2.5 * 18176 * 18176 * 8))    # a capped context runs ONE batch at a time within the cap (no second workspace on a helper): split
                                                           # down to what fits, still the reference's rows, and never more than the cap held
    st, zs = fa.linkage_batch(probs, ctx=ctx)
    assert list(st) == [0] * 6
    for z, zr in zip(zs, singles):
        np.testing.assert_array_equal(z, zr)
    assert ctx.workspace_bytes() <= int(2.5 * 18176 * 18176 * 8) + 6 * probs[2].nbytes + (1 << 20)
    gpu_ctx.trim()
    torch.cuda.empty_cache()


def test_random_batches_through_whichever_path_serves_them(fa, gpu_ctx, oracle_mod):
    """Seeded random batches — 2 .. 6 problems, 2 .. 2 400 points each, both distributions, duplicated rows now and then: whichever path the
    dispatcher picks (uniform layout when the padded sizes lie within a factor of two, the block-map / kernel-argument kernels otherwise, the
    single-launch kernel for one-block problems, reference order on exact ties), every dendrogram is the reference build's, row for row."""
    rng = np.random.default_rng(2024)
    for trial in range(8):
        k = int(rng.integers(2, 7))
        base = int(rng.integers(2, 2400))
        probs = []
        for j in range(k):
            n = max(2, int(base * rng.uniform(0.45, 1.0))) if trial % 2 == 0 else int(rng.integers(2, 2400))
            d = 48
            if rng.random() < 0.5:
                x = speaker_mixture(n, d, int(rng.integers(2, 12)), float(rng.uniform(0.02, 0.08)), int(rng.integers(1 << 30)))
            else:
                x = oracle_mod.ahc_normalize(rng.standard_normal((n, d)))
            if rng.random() < 0.2 and n > 10:
                x = x.copy()
                x[n // 2] = x[0]                                   # one exact tie (distance 0): the reference's heap order decides
            probs.append(x)
        mode = fa.AHC_MODE_AUTO
        st, zs = fa.linkage_batch(probs, mode=mode, ctx=gpu_ctx)
        assert list(st) == [0] * k, (trial, st, [len(p) for p in probs])
        for x, z in zip(probs, zs):
            sr, zr = oracle_mod.linkage_ref(x)
            assert sr == 0
            np.testing.assert_array_equal(z, zr, err_msg=f"trial {trial}, sizes {[len(p) for p in probs]}")


@pytest.mark.gpu
@pytest.mark.parametrize("cpt", ["1", "2", "4"])
def test_slots_per_thread_forms_equal_reference_build(fa, gpu_ctx, oracle_mod, switch, cpt):
    """Round 5: a thread of the round kernel may own 1, 2 or 4 consecutive slots (ahc_round_body's CPT; a block record then covers 256 x CPT slots).
    Which form serves a call is a matter of speed (one slot per thread for a chain of its own, two where a launch holds many workgroups or where
    that makes a short recording one block); every form must give the reference build's dendrogram bit for bit — single problems through the
    multi-block and the single-block kernels in both modes, exact ties through the reference-order route, a NaN as status 5, and uniform batches."""
    rng = np.random.default_rng(int(cpt))
    cases = [oracle_mod.ahc_normalize(rng.standard_normal((n, d))) for n, d in ((2, 3), (3, 5), (257, 16), (513, 32), (700, 64), (1024, 8), (1500, 24))]
    cases.append(speaker_mixture(2300, 32, 12, 0.03, 7))
    dup = oracle_mod.ahc_normalize(rng.standard_normal((400, 8)))
    dup = np.concatenate([dup, dup[:150]])                     # exact ties at the minimum
    for single_block in (True, False):
        switch("FA_AHC_CPT", cpt)
        if single_block:
            switch("FA_AHC_NO_SINGLE_BLOCK", None)
        else:
            switch("FA_AHC_NO_SINGLE_BLOCK", "1")
        for mode in (fa.AHC_MODE_AUTO, fa.AHC_MODE_EXACT):
            for x in cases:
                st, z = fa.linkage(x, ctx=gpu_ctx, mode=mode)
                sr, zr = oracle_mod.linkage_ref(x)
                assert st == sr == 0
                np.testing.assert_array_equal(z, zr)
        st, z, stats = fa.linkage(dup, ctx=gpu_ctx, return_stats=True)
        assert st == 0 and stats["reference_order"] == 1
        np.testing.assert_array_equal(z, oracle_mod.linkage_ref(dup)[1])
        bad = cases[4].copy()
        bad[300, 5] = np.nan
        assert fa.linkage(bad, ctx=gpu_ctx)[0] == 5
    switch("FA_AHC_CPT", None)
    switch("FA_AHC_NO_SINGLE_BLOCK", None)
    switch("FA_AHC_UNI_CPT", cpt)
    base = cases[-1]
    tied = base[:2000].copy()
    tied[1000:2000] = tied[0:1000]
    nan = base[:1900].copy()
    nan[11, 2] = np.nan
    group = [base, base[:2100], base[:1800], tied, base[:1300], nan]
    for mode in (fa.AHC_MODE_AUTO, fa.AHC_MODE_EXACT):
        st, zs, stats = fa.linkage_batch(group, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st[:5] == [0] * 5 and st[5] == 5, st
        for k, (x, z) in enumerate(zip(group[:5], zs[:5])):
            zr = oracle_mod.linkage_ref(x)[1]
            if k == 3 and mode == fa.AHC_MODE_EXACT:             # exact mode keeps its own order among exact ties: heights only
                np.testing.assert_array_equal(np.sort(z[:, 2]), np.sort(zr[:, 2]))
                continue
            np.testing.assert_array_equal(z, zr)
        if mode == fa.AHC_MODE_AUTO:
            assert stats[3]["reference_order"] == 1 and stats[0]["reference_order"] == 0


@pytest.mark.parametrize("n", [65536, 65600])
def test_register_path_boundary_equals_the_reference_build(fa, gpu_ctx, oracle_mod, n):
    """65 536 points is the largest problem whose first reduction holds its block records in registers (four per lane) and reads them PACKED — slot, neighbour
    slot + 1 and both node ids at the full width of their bit fields (round 6, rec_pack); 65 600 is the first size of the many-record kernel, which keeps
    the unpacked arrays.  d = 2 keeps the reference build at ~12 s on one core; the dendrogram must be its, row for row."""
    x = np.random.default_rng(n).standard_normal((n, 2))
    sr, zr = oracle_mod.linkage_ref(x)
    st, z, stats = fa.linkage(x, ctx=gpu_ctx, return_stats=True)
    gpu_ctx.trim()                                            # 34 GB of matrix
    assert st == sr == 0, gpu_ctx.last_error()
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"first differing row {bad[0]} of {n - 1}: device {z[bad[0]]} reference {zr[bad[0]]} ({stats})"
    assert stats["merges"] == n - 1 and stats["reference_order"] == 0
