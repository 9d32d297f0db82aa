"""The "reference order" selection logic (fluidaudio_amd/csrc/ahc_reforder.h: heap, active list, merge bookkeeping — the same header
the HIP kernels compile) replayed on the CPU against the reference build (oracle/_ref) on inputs FULL of exact ties: duplicated rows,
mirrored copies, grid-quantised rows (overlapping tied pairs), regular lattices.  Row for row, bit for bit — this is what fixes the
order among exactly tied distances (fastcluster_internal.hpp:778-935, :1625-1800) that the tie-free device path cannot know."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "cpu", "libahc_reforder_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "cpu", "ahc_reforder_emul.cpp")], check=True)
    lib = C.CDLL(so)
    lib.fa_reforder_emul.argtypes = [np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]

    def run(x):
        x = np.ascontiguousarray(x, np.float64)
        z = np.zeros((len(x) - 1, 4))
        assert lib.fa_reforder_emul(x, x.shape[0], x.shape[1], z) == 0
        return z
    return run


def tie_inputs():
    rng = np.random.default_rng(0)
    base = rng.standard_normal((60, 8))
    yield "duplicates", np.repeat(base, 3, axis=0)[rng.permutation(180)]
    yield "mirrored", np.vstack([np.hstack([base, np.zeros((60, 1))]), np.hstack([-base, np.full((60, 1), 64.0)])])
    yield "grid 1/4", np.round(rng.standard_normal((300, 3)) * 4) / 4
    yield "grid 1/64 clustered", np.round((rng.standard_normal((5, 6))[rng.integers(0, 5, 400)] + 0.08 * rng.standard_normal((400, 6))) * 64) / 64
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(6.0), np.arange(5.0)), -1).reshape(-1, 3)
    yield "lattice", g
    yield "lattice shuffled", g[rng.permutation(len(g))]
    yield "two points", np.array([[0.0, 1.0], [1.0, 0.0]])
    yield "identity 8 twice", np.vstack([np.eye(8), np.eye(8)])
    yield "tie-free", rng.standard_normal((500, 16))
    yield "all equal", np.ones((40, 4))


@pytest.mark.parametrize("name,x", list(tie_inputs()), ids=[n for n, _ in tie_inputs()])
def test_selection_logic_reproduces_the_reference_row_for_row(oracle_mod, emul, name, x):
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    x = np.ascontiguousarray(x, np.float64)
    st, zr = oracle_mod.linkage_ref(x)
    assert st == 0
    z = emul(x)
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"{name}: first differing row {bad[0]} of {len(z)}: emulation {z[bad[0]]} reference {zr[bad[0]]}"


@pytest.mark.parametrize("seed", range(24))
def test_selection_logic_on_random_tie_heavy_inputs(oracle_mod, emul, seed):
    """Randomised: size, dimension, grid step, duplication rate and a mirror are drawn per seed — low-entropy inputs whose distance
    multisets are full of exact ties (often overlapping ones).  Row for row against the reference build."""
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(1000 + seed)
    n, d = int(rng.integers(3, 260)), int(rng.integers(1, 7))
    step = float(rng.choice([1.0, 0.5, 0.25, 1 / 3, 1 / 64]))
    x = np.round(rng.standard_normal((n, d)) * rng.choice([1.0, 2.0, 4.0]) / step) * step
    if rng.random() < 0.5:                                   # duplicated rows
        k = max(1, n // 3)
        x[rng.integers(0, n, k)] = x[rng.integers(0, n, k)]
    if rng.random() < 0.3:                                   # a mirrored copy far away: every intra-copy distance occurs twice
        x = np.vstack([x, -x + 50.0])
    x = np.ascontiguousarray(x[rng.permutation(len(x))], np.float64)
    st, zr = oracle_mod.linkage_ref(x)
    assert st == 0
    z = emul(x)
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"seed {seed} (n {len(x)}, d {d}, step {step}): first differing row {bad[0]}: emulation {z[bad[0]]} reference {zr[bad[0]]}"
