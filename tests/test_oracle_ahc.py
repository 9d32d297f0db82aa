"""Pins the AHC oracle: the reference's own C++ (oracle/_ref) + the Swift pre/post restatement,
against Tests/FluidAudioTests/Diarizer/Offline/AHCClusteringTests.swift:12-145 and SURVEY.md §8c."""
import ctypes as C
import os

import numpy as np
import pytest
from conftest import same_partition, speaker_mixture

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref",
                                                                "libfastcluster_ref.so")) and
                                not os.path.exists("/root/reference"), reason="reference build unavailable")


def test_reference_probe_rows(oracle_mod):
    # 6-point orthogonal-groups case, rows printed by the reference C++
    g = np.array([[1, 0, 0], [.9, .1, 0], [.95, .05, 0], [0, 1, 0], [0, .9, .1], [0, .95, .05]], float)
    st, z = oracle_mod.linkage_ref(oracle_mod.ahc_normalize(g))
    assert st == 0
    exp = [(0, 2, 0.0526, 2), (3, 5, 0.0526, 2), (4, 7, 0.0843, 3), (1, 6, 0.0843, 3), (8, 9, 1.3739, 6)]
    for row, e in zip(z, exp):
        assert (row[0], row[1], row[3]) == (e[0], e[1], e[3]) and abs(row[2] - e[2]) < 5e-5


def test_swift_known_answers(oracle_mod):
    o = oracle_mod
    assert o.ahc_cluster(np.zeros((0, 3)), 0.7).size == 0
    assert o.ahc_cluster([[1.0, 0.0, 0.0]], 0.7).tolist() == [0]
    assert len(set(o.ahc_cluster([[1.0, 2.0, 3.0]] * 5, 0.7).tolist())) == 1
    g = [[1, 0, 0], [.9, .1, 0], [.95, .05, 0], [0, 1, 0], [0, .9, .1], [0, .95, .05]]
    r = o.ahc_cluster(g, 0.8).tolist()
    assert len(set(r[:3])) == 1 and len(set(r[3:])) == 1 and r[0] != r[3]
    e4 = [[1, 0, 0], [.9, .1, 0], [0, 1, 0], [0, .9, .1]]
    assert len(set(o.ahc_cluster(e4, 0.5).tolist())) == 2 and len(set(o.ahc_cluster(e4, 1.5).tolist())) == 1
    eye = np.eye(3)
    assert sorted(set(o.ahc_cluster(eye, 0.5).tolist())) == [0, 1, 2]
    assert len(set(o.ahc_cluster(eye, 2.0).tolist())) == 1
    assert len(set(o.ahc_cluster(eye, 0.0).tolist())) == 3
    assert o.ahc_cluster(np.zeros((3, 0)), 0.7).tolist() == [0, 0, 0]
    assert len(set(o.ahc_cluster(eye, float("nan")).tolist())) == 3  # NaN threshold -> 0 (:112-116)


def test_status_contract_of_reference_build(oracle_mod):
    ref = oracle_mod.ref().fastcluster_compute_centroid_linkage
    x = np.zeros((3, 2))
    z = np.zeros(8)
    assert ref(None, 3, 2, z.ctypes.data, 8) == 1
    assert ref(x.ctypes.data, 0, 2, z.ctypes.data, 8) == 0
    assert ref(x.ctypes.data, 3, 0, z.ctypes.data, 8) == 1
    assert ref(x.ctypes.data, 3, 2, z.ctypes.data, 7) == 3
    assert ref(x.ctypes.data, 1, 2, z.ctypes.data, 0) == 0
    x[1, 0] = np.nan
    assert ref(x.ctypes.data, 3, 2, z.ctypes.data, 8) == 5


def test_naive_restatement_equals_reference_build(oracle_mod):
    rng = np.random.default_rng(0)
    for n, d in ((40, 8), (150, 32), (300, 64)):
        x = oracle_mod.ahc_normalize(rng.standard_normal((n, d)))
        s1, a = oracle_mod.linkage_ref(x)
        s2, b = oracle_mod.linkage_naive(x)
        assert s1 == s2 == 0
        np.testing.assert_array_equal(a, b)  # bit-exact on tie-free data


def test_gpu_algorithm_model_equals_reference_build(oracle_mod):
    """oracle/ahc_model.c replays the round structure of csrc/ahc_round_body.h on the CPU: both the exact-row mode and the
    Lance-Williams + exact-verify mode must reproduce the reference dendrogram bit for bit."""
    import subprocess
    here = os.path.join(os.path.dirname(__file__), "..", "oracle")
    so = os.path.join(here, "libahc_model.so")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(here, "ahc_model.c"), "-lm"], check=True)
    lib = C.CDLL(so)

    class St(C.Structure):
        _fields_ = [(k, C.c_long) for k in ("merges", "rescans", "rounds", "ambiguous", "exact_evals")]

    for x in (oracle_mod.ahc_normalize(np.random.default_rng(1).standard_normal((700, 48))), speaker_mixture(800, 64, 16, 0.03, 2)):
        _, zr = oracle_mod.linkage_ref(x)
        # (mode, eps_scale): exact rows; Lance-Williams filter with the library's bound; the same with the bound
        # inflated 1e8x so that almost every step goes through the WINDOW (exact re-evaluation) round
        for mode, eps_scale in ((0, 0.0), (1, 16.0), (1, 1.6e9)):
            z = np.zeros_like(zr)
            st = St()
            rc = lib.ahc_model_linkage(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]),
                                       z.ctypes.data_as(C.c_void_p), mode, C.c_double(eps_scale), C.byref(st))
            assert rc == 0 and st.merges == x.shape[0] - 1
            np.testing.assert_array_equal(z, zr)
            if eps_scale > 1e6:
                assert st.ambiguous > 100
        # round-2 row bookkeeping (second-minimum bound e2, row copies of pairs that existed at the start-up): the same dendrogram
        # with (almost) no forced re-scans — on both distributions every point's nearest neighbour is the growing cluster
        lib.ahc_model_set_second_bound(0)
        base = St()
        z0 = np.zeros_like(zr)
        lib.ahc_model_linkage(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]), z0.ctypes.data_as(C.c_void_p), 1, C.c_double(16.0), C.byref(base))
        lib.ahc_model_set_second_bound(1)
        for mode, eps_scale in ((0, 0.0), (1, 16.0), (1, 1.6e9)):
            z = np.zeros_like(zr)
            st = St()
            rc = lib.ahc_model_linkage(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]),
                                       z.ctypes.data_as(C.c_void_p), mode, C.c_double(eps_scale), C.byref(st))
            assert rc == 0 and st.merges == x.shape[0] - 1
            np.testing.assert_array_equal(z, zr)
            if (mode, eps_scale) == (1, 16.0):
                assert st.rescans * 10 <= base.rescans and base.rescans > 50, (st.rescans, base.rescans)
        lib.ahc_model_set_second_bound(0)


def test_cut_is_top_down_not_fcluster(oracle_mod):
    # inversion: child higher than parent; the cut stops at the first node <= thr (AHCClustering.swift:155-188)
    z = np.array([[0, 1, 1.0, 2], [2, 3, 0.5, 3]], float)  # node 3 (h=1.0) under root node 4 (h=0.5)
    assert oracle_mod.ahc_cut(z, 3, 0.7).tolist() == [0, 0, 0]
    assert oracle_mod.ahc_cut(z, 3, 0.4).tolist() == [0, 1, 2]


def test_committed_golden_matches_reference_build(oracle_mod):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ahc_golden.npz"))
    for x, z, lab, thr in ((g["xm"], g["zm"], g["labels_m"], 0.6), (g["xi"], g["zi"], g["labels_i"], 1.2)):
        st, zz = oracle_mod.linkage_ref(x)
        assert st == 0
        np.testing.assert_array_equal(zz, z)
        assert same_partition(oracle_mod.ahc_cut(zz, x.shape[0], thr), lab)
    assert len(set(g["labels_m"].tolist())) == 12  # the mixture recovers its 12 speakers at thr 0.6


def _merge_sets(z, n):
    """{frozenset(leaves of the merged cluster): height} of a SciPy-style linkage matrix — independent of the order of its rows."""
    mem = [frozenset([i]) for i in range(n)]
    out = {}
    for a, b, h, s in z:
        m = mem[int(a)] | mem[int(b)]
        assert len(m) == int(s)
        mem.append(m)
        out[m] = float(h)
    return out


def test_scipy_centroid_linkage_is_the_same_tree(oracle_mod):
    """Second opinion: scipy.cluster.hierarchy.linkage(method="centroid") is another implementation of greedy centroid
    linkage (Lance-Williams updates on the condensed distance matrix, rows SORTED by height afterwards; the reference keeps centroids and
    emits rows in merge order, FastClusterWrapper.cpp:169-192).  On tie-free input both describe the same tree: the same clusters are
    formed, at the same heights up to the rounding of the two formulations.  SciPy relabels clusters after sorting, so the comparison is
    over the SETS of leaves each merge creates, not over row contents."""
    from scipy.cluster.hierarchy import linkage
    rng = np.random.default_rng(11)
    for n, d, mix in ((60, 8, False), (250, 32, False), (400, 64, True)):
        x = speaker_mixture(n, d, 9, 0.05, seed=n) if mix else rng.standard_normal((n, d))
        x = oracle_mod.ahc_normalize(x)
        st, z = oracle_mod.linkage_ref(x)
        assert st == 0
        a, b = _merge_sets(z, n), _merge_sets(linkage(x, method="centroid", metric="euclidean"), n)
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, a[k]), (len(k), a[k], b[k])
