import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"
# The library reads its environment ONCE, when it is first used; this asks it to let the test hooks act (fa_debug_inject_fault,
# fa_debug_set_switch: include/fluidaudio_hip.h).  Child processes the tests start (the C hosts) inherit it.
os.environ["FLUIDAUDIO_HIP_DEBUG_HOOKS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def fa():
    """The product package with its HIP library built (cross-compiles without a GPU)."""
    import fluidaudio_amd
    if not os.path.exists(fluidaudio_amd._lib.LIB_PATH):
        fluidaudio_amd.build()
    return fluidaudio_amd


@pytest.fixture
def switch(fa):
    """switch(name, value): force one route of the dispatch for the rest of the test (fa_debug_set_switch — the library does not look at the
    environment after start-up); what a test set is unset again afterwards."""
    touched = []

    def set_switch(name, value="1"):
        st = fa.lib().fa_debug_set_switch(name.encode(), None if value is None else str(value).encode())
        assert st == 0, f"fa_debug_set_switch({name}) -> {st}"
        touched.append(name)
    yield set_switch
    for name in touched:
        fa.lib().fa_debug_set_switch(name.encode(), None)


@pytest.fixture(scope="session")
def gpu_ctx(fa):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test selected but no GPU is visible (there is no CPU fallback)")
    return fa.default_context(0)


def synth_audio(n, seed=1234, scale=0.1):
    """SURVEY.md §8d config-2 style signal: U(-1,1)*scale + two sinusoids."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / 16000.0
    x = rng.uniform(-1, 1, n) * scale + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)
    return x.astype(np.float32)


def speaker_mixture(n, d=256, k=64, sigma=0.02, seed=0):
    """K-speaker mixture, unit rows."""
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((k, d))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    x = c[np.arange(n) % k] + sigma * rng.standard_normal((n, d))
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def same_partition(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    fwd, bwd = {}, {}
    for x, y in zip(a.tolist(), b.tolist()):
        if fwd.setdefault(x, y) != y or bwd.setdefault(y, x) != x:
            return False
    return True
