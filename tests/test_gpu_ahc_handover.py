"""AUTO's tie route hands the problem back to the filter-based rounds once the ties have stopped (round 6; csrc/ahc_rom.hip, prob_adopt in csrc/ahc_rounds.hip).
Ties at distance 0 only (duplicated rows): the rows behind the last tie have a unique closest pair, so the rounds produce what the reference's heap produces.
Whatever happens — handed over, handed over and a later tie met (everything again in reference order), never handed over — the dendrogram is the reference
build's (oracle/_ref) row for row."""
import numpy as np
import pytest

from conftest import speaker_mixture

pytestmark = pytest.mark.gpu


def _duplicated(n, d, dup, seed):
    rng = np.random.default_rng(seed)
    x = speaker_mixture(n, d, 12, 0.05, seed)
    k = int(dup * n)
    x[rng.integers(0, n, k)] = x[rng.integers(0, n, k)]
    return np.ascontiguousarray(x)


@pytest.mark.parametrize("n,d,dup", [(9000, 32, 0.3), (12000, 16, 0.1)])
def test_duplicates_are_handed_over_and_equal_the_reference(fa, gpu_ctx, oracle_mod, switch, n, d, dup):
    x = _duplicated(n, d, dup, 3 * n + d)
    sr, zr = oracle_mod.linkage_ref(x)
    st, z, stats = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)
    assert st == sr == 0, gpu_ctx.last_error()
    assert stats["reference_order"] == 1 and 0 < stats["handed_over_at"] < n - 1 - 4096, stats      # the tie route was taken, and left again
    assert int((zr[: stats["handed_over_at"], 2] == 0).sum()) > 0                                      # (the ties were duplicates)
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"first differing row {bad[0]} (handed over at {stats['handed_over_at']}): device {z[bad[0]]} reference {zr[bad[0]]}"
    # the same problem with the hand-over switched off, and in the mode that never hands over: the same rows
    switch("FA_AHC_RO_NO_HANDOVER", "1")
    st2, z2, stats2 = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)
    assert st2 == 0 and stats2["handed_over_at"] == 0 and stats2["reference_order"] == 1
    np.testing.assert_array_equal(z2, zr)
    switch("FA_AHC_RO_NO_HANDOVER", None)
    st3, z3, stats3 = fa.linkage(x, mode=fa.AHC_MODE_REFERENCE_ORDER, ctx=gpu_ctx, return_stats=True)
    assert st3 == 0 and stats3["handed_over_at"] == 0
    np.testing.assert_array_equal(z3, zr)


def test_a_tie_behind_the_hand_over_sends_the_problem_back(fa, gpu_ctx, oracle_mod):
    """Duplicates first, then — long after the ties have stopped — two far-away pairs at EXACTLY the same distance: the rounds that adopted the problem halt on the
    exact tie at the minimum, and the whole problem runs again in reference order (handed_over_at == -1)."""
    n, d = 9000, 32
    x = _duplicated(n, d, 0.25, 77)
    x = np.concatenate([x, np.zeros((n, 1))], axis=1)                      # one more coordinate keeps the four extra points away from the data
    extra = np.zeros((4, d + 1))
    extra[:, d] = [50.0, 50.0, -50.0, -50.0]
    extra[1, 0] = 0.5                                                      # |e0 - e1|^2 = |e2 - e3|^2 = 0.25, exactly
    extra[3, 0] = 0.5
    x = np.ascontiguousarray(np.concatenate([x, extra]))
    sr, zr = oracle_mod.linkage_ref(x)
    at = np.nonzero(zr[:, 2] == 0.5)[0]
    assert at.size == 2 and at[1] == at[0] + 1 and at[0] > 6000, at        # the tied pairs merge one after the other, late
    st, z, stats = fa.linkage(x, mode=fa.AHC_MODE_AUTO, ctx=gpu_ctx, return_stats=True)
    assert st == sr == 0, gpu_ctx.last_error()
    assert stats["reference_order"] == 1 and stats["handed_over_at"] == -1, stats
    np.testing.assert_array_equal(z, zr)


def test_batch_problems_on_the_tie_route_hand_over_too(fa, gpu_ctx, oracle_mod):
    probs = [_duplicated(9000, 16, 0.2, 5), speaker_mixture(3000, 16, 6, 0.05, 8), _duplicated(12000, 16, 0.2, 6)]
    st, zs, stats = fa.linkage_batch(probs, ctx=gpu_ctx, return_stats=True)
    assert st == [0, 0, 0]
    for x, z in zip(probs, zs):
        np.testing.assert_array_equal(z, oracle_mod.linkage_ref(x)[1])
    assert stats[0]["handed_over_at"] > 0 and stats[2]["handed_over_at"] > 0 and stats[1]["reference_order"] == 0
