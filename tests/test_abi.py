"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("fluidaudio_hip.h", "FastClusterWrapper.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(fa_[a-z0-9_]+|fastcluster_compute_centroid_linkage)\s*\(", text))
    return names


def test_header_and_library_agree(fa):
    lib = fa.lib()
    decl = declared_symbols()
    assert decl == set(fa._lib.EXPORTED_SYMBOLS), decl ^ set(fa._lib.EXPORTED_SYMBOLS)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert b"gfx950" in lib.fa_version()


def test_integration_lists_every_symbol():
    """INTEGRATION.md section 5 names every declared entry next to the reference seam it serves."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    index = text[text.index("## 5. Symbol index"):]
    missing = sorted(n for n in declared_symbols() if f"`{n}`" not in index)
    assert not missing, missing


def test_status_enum_matches_reference_numbering():
    text = open(os.path.join(ROOT, "include", "FastClusterWrapper.h")).read()
    pairs = dict(re.findall(r"FASTCLUSTER_WRAPPER_(\w+)\s*=\s*(\d+)", text))
    # Sources/FastClusterWrapper/include/FastClusterWrapper.h:11-19
    assert pairs == {"SUCCESS": "0", "INVALID_ARGUMENT": "1", "INDEX_OVERFLOW": "2", "OUTPUT_TOO_SMALL": "3",
                     "ALLOCATION_FAILURE": "4", "RUNTIME_ERROR": "5", "UNKNOWN_ERROR": "255"}


def test_linkage_argument_contract_needs_no_gpu(fa):
    """FastClusterWrapper.cpp:203-226: argument errors and trivial sizes are decided before any device work."""
    f = fa.lib().fastcluster_compute_centroid_linkage
    x = np.zeros((3, 2))
    z = np.zeros(8)
    assert f(None, 3, 2, z.ctypes.data, 8) == 1
    assert f(x.ctypes.data, 3, 2, None, 8) == 1
    assert f(x.ctypes.data, 0, 2, z.ctypes.data, 8) == 0
    assert f(x.ctypes.data, 3, 0, z.ctypes.data, 8) == 1
    assert f(x.ctypes.data, 2 ** 31, 2, z.ctypes.data, 8) == 2
    assert f(x.ctypes.data, 3, 2, z.ctypes.data, 7) == 3
    assert f(x.ctypes.data, 1, 2, z.ctypes.data, 0) == 0


def test_product_path_fails_loudly_without_gpu(fa):
    import torch
    if torch.cuda.is_available():
        return
    try:
        fa.Context(0)
    except fa.FluidAudioHipError as e:
        assert e.status == 5
    else:
        raise AssertionError("context creation must fail without a GPU (no CPU fallback)")
    # the drop-in symbol reports RUNTIME_ERROR (status 5) instead of computing on the CPU
    x = np.random.default_rng(0).standard_normal((4, 3))
    z = np.zeros(12)
    assert fa.lib().fastcluster_compute_centroid_linkage(x.ctypes.data, 4, 3, z.ctypes.data, 12) == 5


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "fluidaudio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "fa_oracle" not in text and "oracle/" not in text.replace("oracle/ is test", ""), f


def test_every_entry_survives_null_and_zero_arguments(fa):
    """No entry may crash on NULL pointers / zero sizes — the reference's wrapper answers them with a status
    (FastClusterWrapper.cpp:203-226) and nothing else may cross the ABI.  Runs in a child process so that a crash is a test failure
    naming the entry, not the end of the test run."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import fluidaudio_amd as fa
lib = fa.lib()
for name in sorted(fa._lib.EXPORTED_SYMBOLS):
    f = getattr(lib, name)
    args = []
    for t in (f.argtypes or []):
        if t in (C.c_float, C.c_double):
            args.append(0.0)
        elif t in (C.c_void_p, C.c_char_p) or isinstance(t, type) and issubclass(t, C._Pointer):
            args.append(None)
        else:
            args.append(0)
    print(name, flush=True)
    f(*args)
print("done", flush=True)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "done", f"crashed in {lines[-1] if lines else '?'}: rc {r.returncode} {r.stderr[-300:]}"
    assert len(lines) == len(fa._lib.EXPORTED_SYMBOLS) + 1
