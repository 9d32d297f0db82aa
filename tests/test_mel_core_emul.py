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
