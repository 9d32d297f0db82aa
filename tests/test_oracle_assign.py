"""Known answers of HungarianAssignment / ConstrainedClusterAssignment restated from the reference's XCTest files
(Tests/FluidAudioTests/Diarizer/HungarianAssignmentTests.swift:9-71,
 Tests/FluidAudioTests/Diarizer/Offline/ConstrainedClusterAssignmentTests.swift:8-63) against the CPU oracle."""
import itertools

import numpy as np

NAN = float("nan")
HUNGARIAN = [([[0.9, 0.1], [0.8, 0.2]], [0, 1]), ([[0.1, 0.9, 0.3]], [1]), ([[0.9], [0.5], [0.7]], [0, -1, -1]),
             ([[NAN, 0.2], [0.6, 0.5]], [1, 0])]
CONSTRAINED = [([[0.9, 0.3], [0.8, 0.6]], [0, 0], [0, 1]), ([[0.9, 0.3], [0.8, 0.6]], [0, 1], [0, 0]),
               ([[0.9], [0.2]], [0, 0], [0, -2]), ([[0.1, 0.7, 0.4], [0.5, 0.2, 0.9]], [3, 7], [1, 2])]


def test_hungarian_square_solve(oracle_mod):
    assert oracle_mod.hungarian_solve([], 0).tolist() == []
    assert oracle_mod.hungarian_solve([4, 1, 3, 2, 0, 5, 3, 2, 2], 3).tolist() == [1, 0, 2]
    assert oracle_mod.hungarian_solve([1, 2, 0, 10], 2).tolist() == [1, 0]


def test_max_score_assignment(oracle_mod):
    for scores, want in HUNGARIAN:
        assert oracle_mod.max_score_assignment(scores).tolist() == want
    assert oracle_mod.max_score_assignment([]).tolist() == []
    assert oracle_mod.max_score_assignment([[], []]).tolist() == [-1, -1]


def test_constrained_assignment(oracle_mod):
    for scores, chunks, want in CONSTRAINED:
        assert oracle_mod.constrained_assign(scores, chunks).tolist() == want
    assert oracle_mod.constrained_assign(np.zeros((0, 2)), []).tolist() == []


def test_assignment_is_optimal_on_random_chunks(oracle_mod):
    rng = np.random.default_rng(0)
    for rows, cols in ((3, 5), (3, 3), (2, 4), (4, 3)):
        sc = rng.random((rows, cols))
        got = oracle_mod.max_score_assignment(sc)
        best = -1.0
        for perm in itertools.permutations(range(max(rows, cols)), rows):
            tot = sum(sc[r, c] for r, c in enumerate(perm) if c < cols)
            best = max(best, tot)
        tot = sum(sc[r, c] for r, c in enumerate(got) if c >= 0)
        assert abs(tot - best) < 1e-5 and len({c for c in got if c >= 0}) == sum(c >= 0 for c in got)


def test_assignment_total_equals_scipy_optimum_on_larger_matrices(oracle_mod):
    """An independent solver (scipy.optimize.linear_sum_assignment, maximising) reaches the same TOTAL on matrices too large for the
    permutation check above — rectangular both ways, up to the 3-local-speaker x many-centroid shape of a diarization chunk
    (HungarianAssignment.swift:12-60; which of several optimal assignments is returned is the restatement's business, the total is not)."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    for rows, cols in ((12, 12), (3, 40), (40, 3), (30, 30), (7, 19), (25, 8), (1, 9), (64, 64)):
        for _ in range(3):
            sc = rng.random((rows, cols))
            got = oracle_mod.max_score_assignment(sc)
            used = [c for c in got if c >= 0]
            assert len(set(used)) == len(used) == min(rows, cols)
            tot = sum(sc[r, c] for r, c in enumerate(got) if c >= 0)
            ri, ci = linear_sum_assignment(sc, maximize=True)
            assert abs(tot - sc[ri, ci].sum()) < 1e-4 * min(rows, cols), (rows, cols, tot, sc[ri, ci].sum())
