"""The drop-in boundary without Python in the loop: a C program built with gcc against include/*.h and linked to
libfluidaudio_hip.so (the way the reference's SwiftPM target links FastClusterWrapper) — argument contract on the CPU tier,
a real dendrogram against the reference build on the GPU tier."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "fluidaudio_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cabi", "dropin")


def build(fa):
    fa.lib()                                                    # makes sure the library is built
    src = os.path.join(ROOT, "tests", "cabi", "dropin.c")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(LIBDIR, "libfluidaudio_hip.so"))):
        subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", EXE, "-L", LIBDIR, "-lfluidaudio_hip", "-lpthread",
                        "-Wl,-rpath," + LIBDIR], check=True)
    return EXE


def test_c_host_links_and_keeps_the_argument_contract(fa):
    r = subprocess.run([build(fa), "args"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gfx950" in r.stdout and "violations: 0" in r.stdout


@pytest.mark.gpu
def test_c_host_dendrogram_equals_reference_build(fa, oracle_mod):
    r = subprocess.run([build(fa), "run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "status 0"
    got = np.array([[float(v) for v in ln.split()] for ln in lines[1:6]])
    x = np.array([[1.00, 0.00, 0.0], [0.99, 0.00, 0.141067], [0.998614, 0.052631, 0.0],
                  [0.00, 1.00, 0.0], [0.00, 0.985, 0.172], [0.061, 0.998138, 0.0]])
    st, want = oracle_mod.linkage_ref(x)
    assert st == 0
    np.testing.assert_array_equal(got, want)                    # %.17g round-trips doubles
    assert [tuple(int(v) for v in row[[0, 1, 3]]) for row in got] == [(0, 2, 2), (3, 5, 2), (1, 6, 3), (4, 7, 3), (8, 9, 6)]   # tie-free variant of the SURVEY §8(c) probe


@pytest.mark.gpu
def test_c_host_device_set(fa):
    """fa_pool_* from plain C with pthreads: sharded mel == unsharded mel byte for byte; 6 concurrent callers of the context-free
    drop-in symbol (default pool = device 0 twice) each get the dendrogram of a lone call."""
    r = subprocess.run([build(fa), "pool"], capture_output=True, text=True, timeout=300, env=dict(os.environ, FLUIDAUDIO_HIP_DEVICES="0,0"))
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr


HOST = os.path.join(ROOT, "tests", "cabi", "host")


def build_cpp_host(fa):
    fa.lib()
    src = os.path.join(ROOT, "tests", "cabi", "host.cpp")
    deps = [src, os.path.join(ROOT, "include", "fluidaudio.hpp"), os.path.join(ROOT, "include", "fluidaudio_hip.h"), os.path.join(LIBDIR, "libfluidaudio_hip.so")]
    if not os.path.exists(HOST) or os.path.getmtime(HOST) < max(os.path.getmtime(p) for p in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", HOST, "-L", LIBDIR,
                        "-lfluidaudio_hip", "-Wl,-rpath," + LIBDIR], check=True)
    return HOST


def test_cpp_host_mirror_compiles_and_links(fa):
    """include/fluidaudio.hpp (the C++ mirror of the reference's Swift types) builds warning-free; its host-only pieces work."""
    r = subprocess.run([build_cpp_host(fa), "link"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_runs_the_reference_test_cases(fa):
    """tests/cabi/host.cpp: the reference's own XCTest cases restated in C++ against include/fluidaudio.hpp."""
    r = subprocess.run([build_cpp_host(fa)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAIL" not in r.stdout and "0 failed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("PASS") >= 50


@pytest.mark.gpu
def test_c_host_runs_the_whole_clustering_stage(fa, oracle_mod, tmp_path):
    """fa_offline_cluster from a plain C host (no Python between the caller and the library): labels equal the CPU restatement of
    OfflineDiarizerManager.cluster on the same session."""
    import struct
    from test_gpu_pipeline import synth_session
    emb, rho, chunks, phi, _ = synth_session(250, 5, 7)
    path = tmp_path / "session.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<qii", emb.shape[0], emb.shape[1], rho.shape[1]))
        f.write(np.ascontiguousarray(emb, np.float32).tobytes())
        f.write(np.ascontiguousarray(rho, np.float64).tobytes())
        f.write(np.ascontiguousarray(chunks, np.int32).tobytes())
        f.write(np.ascontiguousarray(phi, np.float64).tobytes())
    r = subprocess.run([build(fa), "cluster", str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi)
    assert lines[0].startswith("status 0 clusters %d training %d " % (ref["centroids"].shape[0], emb.shape[0])) and lines[0].endswith("constrained 1 vbx_degraded 0 ahc_degraded 0")
    assert [int(v) for v in lines[1:]] == ref["assignments"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("site,kw,flags", [(0, dict(vbx_fails=True), "vbx_degraded 1 ahc_degraded 0"), (4, dict(ahc_fails=True), "vbx_degraded 0 ahc_degraded 1")])
def test_c_host_clustering_stage_degrades_like_the_reference(fa, oracle_mod, tmp_path, site, kw, flags):
    """Fault injection through the plain C host: a failing VBx leaves gamma = one-hot AHC labels, pi = 1/S, no ELBOs and the stage goes on
    (VBxClustering.swift:136-141); a failing linkage leaves one cluster per training row (AHCClustering.swift:52-55).  Status stays SUCCESS,
    labels equal the CPU restatement of the same degrade."""
    import struct
    from test_gpu_pipeline import synth_session
    emb, rho, chunks, phi, _ = synth_session(40, 4, 11)
    path = tmp_path / "session.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<qii", emb.shape[0], emb.shape[1], rho.shape[1]))
        f.write(np.ascontiguousarray(emb, np.float32).tobytes())
        f.write(np.ascontiguousarray(rho, np.float64).tobytes())
        f.write(np.ascontiguousarray(chunks, np.int32).tobytes())
        f.write(np.ascontiguousarray(phi, np.float64).tobytes())
    r = subprocess.run([build(fa), "cluster", str(path), str(site)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi, **kw)
    assert lines[0].startswith("status 0 clusters %d " % ref["centroids"].shape[0]) and lines[0].endswith(flags), lines[0]
    if site == 0:
        assert " vbx_iterations 0 " in lines[0]
    assert [int(v) for v in lines[1:]] == ref["assignments"].tolist()
