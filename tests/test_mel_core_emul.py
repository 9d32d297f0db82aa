"""Replays the mel kernel's 16-lane dataflow (fluidaudio_amd/csrc/mel_core.h, the same source the GPU compiles)
on the host and compares the 257 power bins with a float64 DFT: checks the radix-16 passes, the LDS transpose
indices and the even/odd recombination without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


import pytest


@pytest.mark.parametrize("entry", ["mel_core_emul_power", "mel_core_emul_power_v2"])
def test_lane_dataflow_matches_float64_dft(oracle_mod, entry):
    so = os.path.join(HERE, "cpu", "libmel_core_emul.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "cpu", "mel_core_emul.cpp")], check=True)
    lib = C.CDLL(so)
    f32 = np.ctypeslib.ndpointer(np.float32)
    fn = getattr(lib, entry)
    fn.argtypes = [f32, f32, f32]
    rng = np.random.default_rng(1)
    for off, win in ((56, 400), (0, 400), (0, 512)):
        wz = np.zeros(512, np.float32)
        wz[off:off + win] = oracle_mod.hann(win)
        for _ in range(4):
            x = (rng.standard_normal(512) * 0.1).astype(np.float32)
            p = np.zeros(257, np.float32)
            fn(x, wz, p)
            ref = np.abs(np.fft.rfft(x.astype(np.float64) * wz.astype(np.float64))) ** 2
            assert np.max(np.abs(p - ref)) < 1e-6 * ref.max()
    # impulse at every position exercises each lane/index path exactly
    wz = np.ones(512, np.float32)
    for pos in (0, 1, 2, 31, 32, 33, 255, 256, 257, 510, 511):
        x = np.zeros(512, np.float32)
        x[pos] = 1.0
        p = np.zeros(257, np.float32)
        fn(x, wz, p)
        np.testing.assert_allclose(p, 1.0, atol=2e-6)


def test_packed_lane_dataflow_matches_float64_dft(oracle_mod):
    """fluidaudio_amd/csrc/mel_pk.h (two frames per lane) replayed on the host with clang: both halves of every register pair
    must carry an independent, correct 512-point power spectrum; the zero-window-edge variant must agree when it applies."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("clang++ (ext_vector_type) not available")
    so = os.path.join(HERE, "cpu", "libmel_pk_emul.so")
    subprocess.run([clang, "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, os.path.join(HERE, "cpu", "mel_pk_emul.cpp")], check=True)
    lib = C.CDLL(so)
    f32 = np.ctypeslib.ndpointer(np.float32)
    lib.mel_pk_emul_power.argtypes = [f32, f32, f32, C.c_int, f32, f32]
    rng = np.random.default_rng(2)
    for off, win in ((56, 400), (0, 400), (0, 512)):
        wz = np.zeros(512, np.float32)
        wz[off:off + win] = oracle_mod.hann(win)
        ez_ok = not wz[:32].any() and not wz[480:].any()
        for ez in ((0, 1) if ez_ok else (0,)):
            for _ in range(3):
                xa = (rng.standard_normal(512) * 0.1).astype(np.float32)
                xb = (rng.standard_normal(512) * 3.0).astype(np.float32)        # different scale: the halves must not mix
                pa, pb = np.zeros(257, np.float32), np.zeros(257, np.float32)
                lib.mel_pk_emul_power(xa, xb, wz, ez, pa, pb)
                for x, p in ((xa, pa), (xb, pb)):
                    ref = np.abs(np.fft.rfft(x.astype(np.float64) * wz.astype(np.float64))) ** 2
                    assert np.max(np.abs(p - ref)) < 1e-6 * ref.max(), (off, win, ez)
    wz = np.ones(512, np.float32)
    for pos in (0, 1, 2, 31, 32, 33, 255, 256, 257, 510, 511):
        xa, xb = np.zeros(512, np.float32), np.zeros(512, np.float32)
        xa[pos] = 1.0
        xb[511 - pos] = 2.0
        pa, pb = np.zeros(257, np.float32), np.zeros(257, np.float32)
        lib.mel_pk_emul_power(xa, xb, wz, 0, pa, pb)
        np.testing.assert_allclose(pa, 1.0, atol=2e-6)
        np.testing.assert_allclose(pb, 4.0, atol=8e-6)
