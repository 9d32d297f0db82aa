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
    lib = _build("ahc_reforder_emul")
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


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Round 5: the matrix-filtered reference-order run (ahc_rom.hip: rom_scan / rom_select) — entries that carry their key, block-wise sifts
# (ahc_reforder.h: HeapK), a Lance-Williams matrix supplying the candidates of every scan.  tests/cpu/ahc_rom_emul.cpp replays both on the CPU.

def _build(name):
    so = os.path.join(HERE, "cpu", f"lib{name}.so")
    tmp = f"{so}.{os.getpid()}.tmp"                           # built aside and renamed: parallel test workers never load a half-written library
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", tmp, os.path.join(HERE, "cpu", f"{name}.cpp")], check=True)
    os.replace(tmp, so)
    return C.CDLL(so)


@pytest.fixture(scope="module")
def rom():
    lib = _build("ahc_rom_emul")
    lib.fa_heapk_equiv.argtypes = [C.c_int, C.c_int, C.c_ulonglong, C.c_int]
    lib.fa_rom_emul.argtypes = [np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_int, C.c_int, C.c_double, C.c_ulonglong,
                                np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]

    def run(x, noise=0.0, seed=0):
        x = np.ascontiguousarray(x, np.float64)
        z, stats = np.zeros((len(x) - 1, 4)), np.zeros(4)
        assert lib.fa_rom_emul(x, x.shape[0], x.shape[1], noise, seed, z, stats) == 0
        return z, stats
    run.lib = lib
    return run


@pytest.mark.parametrize("n,levels", [(2, 1), (3, 2), (7, 1), (64, 3), (127, 2), (128, 50), (510, 3), (511, 2), (512, 4), (1000, 5), (5000, 3), (20000, 1000), (70000, 7), (140000, 3)])
def test_block_heap_equals_the_restated_heap_after_every_operation(rom, n, levels):
    """remove / replace / raise in random order on heavily tied keys: place by place the same arrays (the tie order IS the array order).  Sizes on
    both sides of the block boundaries: 511 places = one sift-down block (eight levels), 140 000 = three blocks deep."""
    for seed in range(6):
        assert rom.lib.fa_heapk_equiv(n, min(4 * n, 6000), seed, levels) == 0, (n, levels, seed)


@pytest.mark.parametrize("noise", [0.0, 0.45])
@pytest.mark.parametrize("name,x", list(tie_inputs()), ids=[n for n, _ in tie_inputs()])
def test_matrix_filtered_run_reproduces_the_reference_row_for_row(oracle_mod, rom, name, x, noise):
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    x = np.ascontiguousarray(x, np.float64)
    st, zr = oracle_mod.linkage_ref(x)
    assert st == 0
    z, stats = rom(x, noise, 7)
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"{name}: first differing row {bad[0]} of {len(z)}: emulation {z[bad[0]]} reference {zr[bad[0]]} (stats {stats})"
    assert stats[1] >= stats[0] >= len(x) - 2


@pytest.mark.parametrize("seed", range(16))
def test_matrix_filtered_run_on_random_inputs(oracle_mod, rom, seed):
    """Tie-heavy and tie-free inputs by turns, the start-up matrix perturbed by up to 0.45 eps per entry: the candidates of a scan then really are
    several, and the exact evaluation has to pick the reference's."""
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(2000 + seed)
    n, d = int(rng.integers(3, 300)), int(rng.integers(1, 40))
    x = rng.standard_normal((n, d))
    if seed % 2 == 0:
        step = float(rng.choice([1.0, 0.5, 0.25, 1 / 64]))
        x = np.round(x * 2 / step) * step
        k = max(1, n // 3)
        x[rng.integers(0, n, k)] = x[rng.integers(0, n, k)]
    else:
        x /= np.linalg.norm(x, axis=1, keepdims=True)       # the pipeline's input: unit rows
    x = np.ascontiguousarray(x, np.float64)
    st, zr = oracle_mod.linkage_ref(x)
    assert st == 0
    z, stats = rom(x, 0.45, seed)
    bad = np.nonzero((z != zr).any(axis=1))[0]
    assert bad.size == 0, f"seed {seed} (n {n}, d {d}): first differing row {bad[0]}: emulation {z[bad[0]]} reference {zr[bad[0]]} (stats {stats})"
