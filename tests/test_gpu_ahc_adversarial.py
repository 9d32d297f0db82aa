"""Adversarial parity of the centroid linkage (HIP, FA_AHC_MODE_AUTO = Lance-Williams filter + exact certification windows, and
FA_AHC_MODE_EXACT) against the reference's own C++ build (oracle/_ref; reference: FastClusterWrapper.cpp:45-52,68-75 distances,
fastcluster_internal.hpp:1625-1800 algorithm).

The AUTO path decides from approximate (Lance-Williams / Gram-form) values only when the decision is certified by a 2 eps window;
everything inside the window is re-evaluated with the reference's exact sum.  These inputs attack exactly that:
  * distances a few ulp apart at EVERY level of the hierarchy (a second copy of the point set with permuted coordinates: the same
    distances in exact arithmetic, different rounding of the sequential sums),
  * pairs planted 1 / 2 / 8 / 64 ulp apart in squared distance,
  * fp32-rounded rows widened to fp64 (the real call path, OfflineDiarizerManager.swift:286),
  * rows quantised to a 1/64 grid (many exact non-zero ties AND near-ties, overlapping candidate pairs),
  * mirrored copies (exact ties between disjoint pairs at every level).
Tie-free inputs must reproduce the reference dendrogram bit for bit.  On EXACT ties the reference's order is an artefact of its
binary heap (fastcluster_internal.hpp:778-890): between disjoint pairs the tree is the same up to row order (heights multiset +
every partition equal); between OVERLAPPING pairs centroid linkage is not reducible and two valid greedy runs can build different
trees — there the device's dendrogram is checked to be a valid greedy centroid linkage (every merged pair is a global minimum of
the reference's distance at its step, heights bit-exact) by an independent numpy replay, and compared with the reference where
they agree.  `windows` / `exact_fallback` are asserted so that a silent filter failure is visible."""
import numpy as np
import pytest
from conftest import same_partition

pytestmark = pytest.mark.gpu
THRS = (0.0, 0.05, 0.2, 0.4, 0.6, 0.9, 1.2, 2.0)


def seq_sqdist(x, y):
    """The reference's distance: sequential fp64 sum of squared differences (FastClusterWrapper.cpp:45-52)."""
    d = np.asarray(x, np.float64) - np.asarray(y, np.float64)
    return float(np.cumsum(d * d)[-1])


def replay_is_valid_greedy(x, z):
    """Independent numpy replay of a dendrogram: at every step the merged pair must be A global minimum of the reference's
    distance among the active clusters, the height its square root, the centroid the reference's weighted mean (:89-100)."""
    n, d = x.shape
    cent = {i: x[i].copy() for i in range(n)}
    size = {i: 1.0 for i in range(n)}
    for s in range(n - 1):
        a, b, h, m = int(z[s, 0]), int(z[s, 1]), z[s, 2], z[s, 3]
        ids = sorted(cent)
        c = np.stack([cent[i] for i in ids])
        diff = c[:, None, :] - c[None, :, :]
        dm = np.cumsum(diff * diff, axis=2)[:, :, -1]
        np.fill_diagonal(dm, np.inf)
        dab = dm[ids.index(a), ids.index(b)]
        if dab != dm.min() or h != np.sqrt(dab) or m != size[a] + size[b]:
            return False, s
        cent[n + s] = (cent[a] * size[a] + cent[b] * size[b]) / (size[a] + size[b])
        size[n + s] = size[a] + size[b]
        del cent[a], cent[b]
    return True, -1


def clustered(n, d, k, sigma, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((k, d))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    return c[rng.integers(0, k, n)] + sigma * rng.standard_normal((n, d))


def replay_sq(x, z):
    """Squared merge distance of every row of a dendrogram, recomputed on the CPU with the reference's arithmetic (sequential sums
    :45-52, weighted-mean centroids :89-100) by following the dendrogram's own merges."""
    n, d = x.shape
    cent = np.zeros((2 * n - 1, d))
    cent[:n] = x
    size = np.ones(2 * n - 1)
    out = np.zeros(n - 1)
    for s in range(n - 1):
        a, b = int(z[s, 0]), int(z[s, 1])
        diff = cent[a] - cent[b]
        out[s] = np.cumsum(diff * diff)[-1]
        cent[n + s] = (cent[a] * size[a] + cent[b] * size[b]) / (size[a] + size[b])
        size[n + s] = size[a] + size[b]
    return out


def valid_greedy_fast(x, z):
    """(ok, step): is the dendrogram a valid greedy centroid linkage under the reference's exact arithmetic — at every step the merged
    pair's squared distance is <= that of EVERY active pair (bitwise comparison of sequential fp64 sums), and the height its square
    root?  Full distance matrix on the CPU, one new row per merge, row minima kept with their argmin; O(n^2 d) start + O(n d) per merge."""
    n, d = x.shape
    tot = 2 * n - 1
    cent = np.zeros((tot, d))
    cent[:n] = x
    size = np.ones(tot)
    D = np.full((tot, tot), np.inf)
    for r0 in range(0, n, 128):
        r1 = min(r0 + 128, n)
        diff = x[r0:r1, None, :] - x[None, :n, :]
        D[r0:r1, :n] = np.cumsum(diff * diff, axis=2)[:, :, -1]
    D[np.arange(n), np.arange(n)] = np.inf
    active = np.zeros(tot, bool)
    active[:n] = True
    rowarg = D.argmin(axis=1)
    rowmin = D[np.arange(tot), rowarg]
    for s in range(n - 1):
        a, b, h = int(z[s, 0]), int(z[s, 1]), z[s, 2]
        if not (active[a] and active[b]):
            return False, s
        dab = D[a, b]
        if dab != rowmin[active].min() or h != np.sqrt(dab):
            return False, s
        c = n + s
        cent[c] = (cent[a] * size[a] + cent[b] * size[b]) / (size[a] + size[b])
        size[c] = size[a] + size[b]
        active[a] = active[b] = False
        D[a, :] = D[:, a] = D[b, :] = D[:, b] = np.inf
        idx = np.nonzero(active)[0]
        if len(idx):
            diff = cent[idx] - cent[c]
            row = np.cumsum(diff * diff, axis=1)[:, -1]
            D[c, idx] = row
            D[idx, c] = row
            k = int(row.argmin())
            rowmin[c], rowarg[c] = row[k], idx[k]
            lost = idx[(rowarg[idx] == a) | (rowarg[idx] == b)]
            if len(lost):
                rowarg[lost] = D[lost].argmin(axis=1)
                rowmin[lost] = D[lost, rowarg[lost]]
            better = row < rowmin[idx]
            rowmin[idx[better]] = row[better]
            rowarg[idx[better]] = c
        active[c] = True
    return True, -1


def tree_signature(z, n, sq):
    """The dendrogram as a SET of nodes (leaf set, squared height, size), independent of row order and node numbering: a random
    64-bit weight per leaf, a node's key = the wrapping sum of its leaves' weights."""
    w = [int(v) for v in np.random.default_rng(12345).integers(0, 2 ** 63, 2 * n - 1, dtype=np.uint64)]
    cnt = [1] * (2 * n - 1)
    for s in range(n - 1):
        a, b = int(z[s, 0]), int(z[s, 1])
        w[n + s] = (w[a] + w[b]) & (2 ** 64 - 1)
        cnt[n + s] = cnt[a] + cnt[b]
    return sorted(zip(w[n:], sq.tolist(), cnt[n:]))


def check_exact(fa, gpu_ctx, oracle_mod, x, want_windows=False, modes=(0, 1)):
    """Device dendrogram == the reference build's, bit for bit — in FA_AHC_MODE_AUTO (what the drop-in symbol runs) and
    FA_AHC_MODE_REFERENCE_ORDER without exception: an exact tie at the minimum re-runs the problem in the reference's selection order
    (csrc/ahc_reforder.h).  FA_AHC_MODE_EXACT keeps its own documented tie order (value, row, column) and has ONE excuse: rows in a
    different ORDER because squared distances were EXACTLY equal (the reference's order among exact ties is its heap layout,
    fastcluster_internal.hpp:778-890).  That excuse is checked, not assumed: the reference run must contain exact ties, the device's dendrogram must be a VALID greedy run under the
    reference's exact arithmetic (valid_greedy_fast: at every step the merged pair is a global minimum, bitwise — a pair merged
    before one that is 1 ulp closer fails), and both must be the same tree as a set of (leaf set, squared height, size) nodes."""
    sr, zr = oracle_mod.linkage_ref(x)
    assert sr == 0
    sq_ref = None
    out = None
    for mode in modes:
        st, z, stats = fa.linkage(x, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st == 0
        bad = np.nonzero((z != zr).any(axis=1))[0]
        if mode != 1:    # AUTO (re-runs in the reference's selection order on an exact tie) and REFERENCE_ORDER: row for row, no excuse
            assert bad.size == 0, f"mode {mode}: first differing merge {bad[0]} of {len(z)}: device {z[bad[0]]} reference {zr[bad[0]]} stats {stats}"
        if bad.size:
            if sq_ref is None:
                sq_ref = replay_sq(x, zr)
                np.testing.assert_array_equal(np.sqrt(sq_ref), zr[:, 2])
                assert len(np.unique(sq_ref)) < len(sq_ref), "no exact ties in the reference run: every row must match"
            sq = replay_sq(x, z)
            first = bad[0]
            msg = f"mode {mode}: first differing merge {first} of {len(z)}: device {z[first]} reference {zr[first]} stats {stats}"
            ok, step = valid_greedy_fast(x, z)
            assert ok, msg + f"; NOT a valid greedy run: step {step} merged a pair that was not a global minimum (or its height is off)"
            assert tree_signature(z, len(x), sq) == tree_signature(zr, len(x), sq_ref), msg + "; valid, but a different tree"
        if mode == 0:
            out = stats
            out["rows_out_of_order_on_exact_ties"] = int(bad.size)
            if want_windows:
                assert stats["windows"] > 0 or stats["exact_fallback"] > 0, stats   # the filter saw the near-ties
    return out


@pytest.mark.parametrize("n,d,seed", [(600, 24, 1), (1500, 64, 2), (900, 256, 3)])
def test_near_ties_at_every_level_permuted_copy(fa, gpu_ctx, oracle_mod, n, d, seed):
    """Second copy of the set with its coordinates permuted, kept apart by an extra coordinate: in exact arithmetic every distance
    inside copy 2 equals its twin in copy 1; in fp64 the sequential sums round differently — twins 0 .. a few ulp apart at every level
    of both sub-trees (where they come out EXACTLY equal the reference's row order is its heap layout: check_exact)."""
    x = clustered(n, d, 9, 0.05, seed)
    perm = np.random.default_rng(seed + 100).permutation(d)
    y = x[:, perm].copy()
    both = np.hstack([np.vstack([x, y]), np.zeros((2 * n, 1))])
    both[n:, d] = 64.0                                      # an extra LAST coordinate keeps the copies apart (their last merge) and adds
                                                            # an exact 0 to every within-copy sum: twins differ by summation order only
    stats = check_exact(fa, gpu_ctx, oracle_mod, both, want_windows=True)
    assert stats["merges"] == 2 * n - 1


def plant(p, i, j, want):
    """Moves the LAST coordinate of point j (equal to point i's there, so the last term of the sequential sum is the only one that
    changes and the sum can take every fp64 value above its base) until seq_sqdist(p[i], p[j]) == want."""
    k = p.shape[1] - 1
    p[j, k] = p[i, k]
    assert seq_sqdist(p[i], p[j]) <= want
    lo, hi = 0.0, 1.0
    for _ in range(400):
        mid = 0.5 * (lo + hi)
        p[j, k] = p[i, k] + mid
        v = seq_sqdist(p[i], p[j])
        if v == want:
            return
        if v < want:
            lo = mid
        else:
            hi = mid
    raise AssertionError("could not plant the distance")


@pytest.mark.parametrize("ulps", [1, 2, 8, 64])
def test_planted_pairs_ulps_apart(fa, gpu_ctx, oracle_mod, ulps):
    """Two disjoint closest pairs whose squared distances differ by exactly `ulps` ulp (either one the smaller), inside 2 000 other
    points: the first merge must be the reference's, and so must everything after it."""
    rng = np.random.default_rng(50 + ulps)
    n, d = 2000, 48
    x = oracle_mod.ahc_normalize(rng.standard_normal((n, d)))
    for sign in (+1, -1):
        p = x.copy()
        p[1] = p[0]; p[1, 3] += 2.0 ** -12                # pair (0, 1) and pair (2, 3): by far the closest pairs, ~2^-24 apart squared
        p[3] = p[2]; p[3, 5] += 2.0 ** -12
        base = max(seq_sqdist(p[0], p[1]), seq_sqdist(p[2], p[3]))
        for _ in range(70):
            base = np.nextafter(base, np.inf)
        want = base
        for _ in range(ulps):
            want = np.nextafter(want, np.inf)
        lo_pair, hi_pair = ((0, 1), (2, 3)) if sign > 0 else ((2, 3), (0, 1))   # which of the two ends up `ulps` ulp below the other
        plant(p, *lo_pair, base)
        plant(p, *hi_pair, want)
        da, db = seq_sqdist(p[0], p[1]), seq_sqdist(p[2], p[3])
        assert abs(da - db) <= ulps * np.spacing(max(da, db)) and (da < db) == (sign > 0) and da != db
        check_exact(fa, gpu_ctx, oracle_mod, p, want_windows=True)


def test_fp32_rows_widened_like_the_swift_caller(fa, gpu_ctx, oracle_mod):
    """OfflineDiarizerManager.swift:286 widens Float embeddings to Double, AHCClustering normalises in fp64 (:70-105): every
    coordinate carries at most 24 significant bits before the normalisation."""
    for seed, (n, d, k, sigma) in enumerate([(4000, 256, 12, 0.03), (2500, 192, 40, 0.02), (3000, 256, 1, 1.0)]):
        x32 = clustered(n, d, k, sigma, 200 + seed).astype(np.float32)
        x = oracle_mod.ahc_normalize(x32.astype(np.float64))
        check_exact(fa, gpu_ctx, oracle_mod, x, modes=(0,))
        lab = fa.AHCClustering(ctx=gpu_ctx).cluster(x32.astype(np.float64), 0.6)
        np.testing.assert_array_equal(np.asarray(lab, np.int32), oracle_mod.ahc_cluster(x32.astype(np.float64), 0.6))


def test_mirrored_copy_exact_ties_between_disjoint_pairs(fa, gpu_ctx, oracle_mod):
    """x and -x (kept apart by an extra coordinate): bit-identical distances in both halves at every level — exact ties, but only
    between DISJOINT pairs, so the reference's heap order changes the row order of the dendrogram and nothing else: same tree, same
    squared distance at every step (check_exact), same partitions."""
    x = clustered(700, 32, 7, 0.05, 11)
    both = np.hstack([np.vstack([x, -x]), np.zeros((1400, 1))])
    both[700:, 32] = 64.0
    stats = check_exact(fa, gpu_ctx, oracle_mod, both)
    sr, zr = oracle_mod.linkage_ref(both)
    for mode in (0, 1):
        st, z = fa.linkage(both, mode=mode, ctx=gpu_ctx)
        assert st == 0
        for thr in THRS:
            assert same_partition(fa.cut(z, len(both), thr), oracle_mod.ahc_cut(zr, len(both), thr)), (thr, mode, stats)


@pytest.mark.parametrize("n,d,k", [(240, 8, 5), (300, 12, 3)])
def test_quantised_rows_overlapping_ties_valid_greedy(fa, gpu_ctx, oracle_mod, n, d, k):
    """Rows on a 1/64 grid: exact non-zero ties between OVERLAPPING pairs.  Whatever order the ties are taken in, the result must be
    a valid greedy centroid linkage with bit-exact heights (independent numpy replay) — for the device AND for the reference; where
    the two trees coincide as multisets of heights the partitions must coincide too."""
    x = np.round(clustered(n, d, k, 0.08, 31 + n) * 64.0) / 64.0
    x = x[np.abs(x).sum(axis=1) > 0]
    sr, zr = oracle_mod.linkage_ref(x)
    assert sr == 0
    ok_ref, _ = replay_is_valid_greedy(x, zr)
    assert ok_ref                                            # the checker accepts the reference's own output
    for mode in (0, 2, 1):
        st, z, stats = fa.linkage(x, mode=mode, ctx=gpu_ctx, return_stats=True)
        assert st == 0
        if mode != 1:   # the reference's own choice among the tied pairs, row for row
            np.testing.assert_array_equal(z, zr)
            assert stats["reference_order"] == 1 or mode == 0
        ok, step = replay_is_valid_greedy(x, z)
        assert ok, (mode, step, stats)
        if np.array_equal(np.sort(z[:, 2]), np.sort(zr[:, 2])):
            for thr in THRS:
                assert same_partition(fa.cut(z, len(x), thr), oracle_mod.ahc_cut(zr, len(x), thr)), (thr, mode)


def test_quantised_then_normalised_rows_at_size(fa, gpu_ctx, oracle_mod):
    """The verdict's case (i): rows quantised to a 1/64 grid BEFORE the normalisation (n = 4 000).  After the fp64 normalisation
    exact ties survive only between identical rows; everything else becomes near-ties a few ulp apart.  Heights multiset and the
    partitions at 8 thresholds equal the reference's (the fp64 normalisation separates everything by far more than the filter's eps)."""
    x = np.round(clustered(4000, 32, 10, 0.06, 77) * 64.0) / 64.0
    x = x[np.abs(x).sum(axis=1) > 0]
    xn = oracle_mod.ahc_normalize(x)
    sr, zr = oracle_mod.linkage_ref(xn)
    assert sr == 0
    st, z, stats = fa.linkage(xn, mode=0, ctx=gpu_ctx, return_stats=True)
    assert st == 0
    np.testing.assert_array_equal(z, zr)
    np.testing.assert_array_equal(np.sort(z[:, 2]), np.sort(zr[:, 2]))
    for thr in THRS:
        assert same_partition(fa.cut(z, len(xn), thr), oracle_mod.ahc_cut(zr, len(xn), thr)), (thr, stats)
    assert stats["merges"] == len(xn) - 1


def test_massive_exact_ties_equal_the_reference(fa, gpu_ctx, oracle_mod):
    """30 % / 90 % of the rows are exact copies of other rows (round 2 compared the two device modes with each other): the window of
    the Lance-Williams filter overflows, the problem is recomputed in the reference's selection order and equals the reference build's
    dendrogram row for row."""
    from conftest import speaker_mixture
    for n, dup in ((3000, 0.3), (4000, 0.9)):
        x = speaker_mixture(n, 64, 12, 0.03, 7).copy()
        rng = np.random.default_rng(1)
        x[rng.integers(0, n, int(n * dup))] = x[rng.integers(0, n, int(n * dup))]
        sr, zr = oracle_mod.linkage_ref(x)
        assert sr == 0
        st, z, stats = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)
        assert st == 0 and stats["exact_fallback"] == 1 and stats["reference_order"] == 1
        np.testing.assert_array_equal(z, zr)              # row for row, the duplicates merged in the reference's order
        zero = int((zr[:, 2] == 0).sum())
        assert int((z[:, 2] == 0).sum()) == zero
        for thr in (1e-9, 0.3, 0.6, 1.0):   # not 0.0: three mutual copies merge as (2x + x) / 3, off x by an ulp — WHICH copy ends up 1e-17 away is tie order
            assert same_partition(fa.cut(z, n, thr), oracle_mod.ahc_cut(zr, n, thr)), (thr, stats)


@pytest.mark.parametrize("n,d,kind", [(2, 3, "iid"), (3, 2, "iid"), (257, 5, "grid"), (1000, 16, "iid"), (1500, 8, "grid"), (700, 4, "dup")])
def test_reference_order_mode_equals_the_reference(fa, gpu_ctx, oracle_mod, n, d, kind):
    """FA_AHC_MODE_REFERENCE_ORDER on its own (what AUTO falls back to): tie-free, grid-quantised and duplicated inputs, tiny and
    multi-block sizes — the reference build's dendrogram row for row; `rounds` counts the scans (one per row + the re-scans)."""
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d))
    if kind == "grid":
        x = np.round(x * 3) / 3
    if kind == "dup":
        x[rng.integers(0, n, n // 2)] = x[rng.integers(0, n, n // 2)]
    sr, zr = oracle_mod.linkage_ref(x)
    st, z, stats = fa.linkage(x, mode=fa.AHC_MODE_REFERENCE_ORDER, ctx=gpu_ctx, return_stats=True)
    assert st == sr == 0
    np.testing.assert_array_equal(z, zr)
    assert stats["reference_order"] == 1 and stats["merges"] == n - 1
    # the drop-in symbol (AUTO) on the same input
    st2, z2 = fa.fastcluster_compute_centroid_linkage(x)
    assert st2 == 0
    np.testing.assert_array_equal(z2, zr)


def test_batch_with_tied_and_tie_free_problems(fa, gpu_ctx, oracle_mod):
    """fa_ahc_linkage_batch: the tied problems of a batch are recomputed in reference order after the others have finished; every
    dendrogram is the reference's."""
    rng = np.random.default_rng(5)
    probs = [rng.standard_normal((300, 8)), np.round(rng.standard_normal((400, 8)) * 2) / 2, rng.standard_normal((500, 8)), np.repeat(rng.standard_normal((100, 8)), 3, axis=0)]
    st, zs, stats = fa.linkage_batch(probs, ctx=gpu_ctx, return_stats=True)
    assert st == [0, 0, 0, 0]
    for x, z in zip(probs, zs):
        np.testing.assert_array_equal(z, oracle_mod.linkage_ref(x)[1])
    assert stats[1]["reference_order"] == 1 and stats[3]["reference_order"] == 1 and stats[0]["reference_order"] == 0


@pytest.mark.parametrize("form", ["matrix-filter", "matrix-free"])
@pytest.mark.parametrize("n,d,kind", [(2, 3, "iid"), (3, 2, "iid"), (40, 4, "equal"), (300, 5, "grid"), (600, 16, "equal"), (900, 300, "unit"), (1500, 8, "grid"),
                                      (2000, 64, "dup90"), (2500, 256, "dup30"), (3000, 256, "unit"), (3000, 48, "lattice")])
def test_both_reference_order_forms_equal_the_reference(fa, gpu_ctx, oracle_mod, switch, n, d, kind, form):
    """FA_AHC_MODE_REFERENCE_ORDER through the matrix filter (round 5: rom_scan / rom_select — Lance-Williams candidates, exact sums of the few, the
    key-carrying block heap) and matrix-free (FA_AHC_RO_NO_MATRIX: O(A d) sums per row, the restated heap): the reference build's dendrogram row for
    row.  "equal" overflows the candidate list of one wavefront (ROM_EXACT rows), d = 300 takes two staging passes per candidate, d = 5 / 8
    the Gram kernel without the LDS-direct loads."""
    rng = np.random.default_rng(7 * n + d)
    x = rng.standard_normal((n, d))
    if kind == "grid":
        x = np.round(x * 3) / 3
    elif kind == "equal":
        x = np.ones((n, d)) * 0.37
    elif kind == "unit":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    elif kind in ("dup30", "dup90"):
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        k = int(n * (0.3 if kind == "dup30" else 0.9))
        x[rng.integers(0, n, k)] = x[rng.integers(0, n, k)]
    elif kind == "lattice":
        x = np.zeros((n, d))
        x[:, :3] = np.stack(np.meshgrid(np.arange(15.0), np.arange(20.0), np.arange(10.0)), -1).reshape(-1, 3)[rng.permutation(3000)[:n]]
    x = np.ascontiguousarray(x)
    if form == "matrix-free":
        switch("FA_AHC_RO_NO_MATRIX", "1")
    sr, zr = oracle_mod.linkage_ref(x)
    st, z, stats = fa.linkage(x, mode=fa.AHC_MODE_REFERENCE_ORDER, ctx=gpu_ctx, return_stats=True)
    assert st == sr == 0, gpu_ctx.last_error()
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"first differing row {bad[0]} of {n - 1}: device {z[bad[0]]} reference {zr[bad[0]]} ({stats})"
    assert stats["reference_order"] == 1 and stats["merges"] == n - 1
    if kind == "equal" and form == "matrix-filter" and n > 200:
        assert stats["rescans"] > 0, stats                  # rows whose candidates overflowed one wavefront were scanned again with exact sums
    st2, z2, stats2 = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)   # and through AUTO -> tie -> reference order
    assert st2 == 0
    np.testing.assert_array_equal(z2, zr)
