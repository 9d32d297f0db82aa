"""GPU: the device set of the C ABI (csrc/pool.hip).  The test box has ONE GPU, so the pools list device 0 twice (two
contexts = two streams): what is checked is the partitioning, the output geometry and the concurrency contract — a sharded
call must give, bit for bit, what the unsharded entry gives, and concurrent callers of the context-free drop-in symbol
(FastClusterWrapper.h:35-41) must each get the reference's dendrogram."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest
from conftest import speaker_mixture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pool2(fa):
    p = fa.Pool([0, 0])
    yield p
    p.close()


def test_pool_basics(fa, pool2):
    assert fa.device_count() >= 1
    assert len(pool2) == 2 and pool2.devices() == [0, 0]
    with pool2.acquire() as (h1, d1), pool2.acquire() as (h2, d2):
        assert h1.value != h2.value and d1 == d2 == 0
        got = []
        t = threading.Thread(target=lambda: got.append(pool2.acquire().__enter__()))   # a third caller waits ...
        t.start()
        time.sleep(0.2)
        assert not got
    t.join(5)                                                                        # ... until a context comes back
    assert got and got[0][0].value in (h1.value, h2.value)
    fa.lib().fa_pool_release(pool2._h, got[0][0])
    with pytest.raises(fa.FluidAudioHipError):
        fa.Pool([0, 99])


@pytest.mark.parametrize("layout", ["mel_major", "frame_major"])
def test_sharded_mel_equals_unsharded(fa, gpu_ctx, pool2, layout):
    import fluidaudio_amd._lib as L
    rng = np.random.default_rng(5)
    lens = [16000, 0, 240000, 399, 1, 52000, 160001, 8000, 240000, 31999]
    utts = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    last = rng.standard_normal(len(lens)).astype(np.float32)
    m = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    cfg = m.config(L.MEL_PAD_CENTER, L.MEL_LAYOUT_MEL_MAJOR if layout == "mel_major" else L.MEL_LAYOUT_FRAME_MAJOR)
    mel, ml = pool2.mel_batch(cfg, utts, last_samples=last)
    assert mel.shape[0] == len(lens) and (mel.shape[1] == 128 if layout == "mel_major" else mel.shape[2] == 128)
    for b, u in enumerate(utts):
        if layout == "mel_major":
            ref, ref_len, _ = m.compute_flat(u, float(last[b]))
            T = ref.size // 128
            got = mel[b, :, :T].ravel() if ref_len else None
        else:
            ref, ref_len, _ = m.compute_flat_transposed(u, float(last[b]))
            T = ref.size // 128
            got = mel[b, :T, :].ravel() if ref_len else None
        assert int(ml[b]) == ref_len
        if ref_len:
            np.testing.assert_array_equal(got, ref)            # same kernel, same utterance -> same bits
    # a pool with more contexts than utterances, and an empty batch
    p3 = fa.Pool([0, 0, 0])
    mel1, ml1 = p3.mel_batch(cfg, utts[:2], last_samples=last[:2])
    assert int(ml1[0]) == int(ml[0]) and int(ml1[1]) == 0
    mel0, ml0 = p3.mel_batch(cfg, [])
    assert mel0.shape[0] == 0
    p3.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_sharded_ctc_equals_unsharded(fa, gpu_ctx, pool2, dtype):
    rng = np.random.default_rng(11)
    B, T, V = 37, 211, 1025
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    x[..., V - 1] += 2.0
    x = x.astype(dtype)
    vf = rng.integers(0, T + 1, B).astype(np.int32)
    ids, lens, fids = pool2.ctc_greedy_batch(x, blank_id=V - 1, valid_frames=vf, return_frame_ids=True)
    rid, rfid = fa.ctc_greedy_ids_batch(x, blank_id=V - 1, valid_frames=vf, return_frame_ids=True, ctx=gpu_ctx)
    for b in range(B):
        assert lens[b] == rid[b].size
        np.testing.assert_array_equal(ids[b, :lens[b]], rid[b])
        np.testing.assert_array_equal(fids[b, :vf[b]], rfid[b, :vf[b]])


def test_linkage_many_equals_reference_build(fa, pool2, oracle_mod):
    rng = np.random.default_rng(2)
    probs = [speaker_mixture(900, 64, 9, 0.04, 1), oracle_mod.ahc_normalize(rng.standard_normal((400, 64))), speaker_mixture(1500, 64, 12, 0.03, 3),
             np.ones((1, 64)), speaker_mixture(257, 64, 5, 0.05, 2), speaker_mixture(1200, 64, 7, 0.05, 9)]
    bad = speaker_mixture(300, 64, 4, 0.05, 5).copy()
    bad[3, 3] = np.nan
    probs.append(bad)
    st, zs = pool2.linkage_many(probs)
    assert st[:-1] == [0] * (len(probs) - 1) and st[-1] == 5
    for x, z in zip(probs[:-1], zs[:-1]):
        if x.shape[0] >= 2:
            sr, zr = oracle_mod.linkage_ref(x)
            assert sr == 0
            np.testing.assert_array_equal(z, zr)


def test_concurrent_callers_of_the_drop_in_symbol():
    """8 host threads call fastcluster_compute_centroid_linkage at once on a default pool of two contexts
    (FLUIDAUDIO_HIP_DEVICES=0,0): every caller gets the dendrogram a lone call produces."""
    code = r'''
import threading, numpy as np, sys
sys.path.insert(0, %r)
import fluidaudio_amd as fa
rng = np.random.default_rng(0)
xs = [rng.standard_normal((300 + 37 * i, 48)) for i in range(8)]
alone = [fa.fastcluster_compute_centroid_linkage(x) for x in xs]
out = [None] * 8
def work(i):
    out[i] = fa.fastcluster_compute_centroid_linkage(xs[i])
for rep in range(3):
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(8):
        assert out[i][0] == 0 and alone[i][0] == 0 and np.array_equal(out[i][1], alone[i][1]), i
print("CONCURRENT_OK")
''' % ROOT
    env = dict(os.environ, FLUIDAUDIO_HIP_DEVICES="0,0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "CONCURRENT_OK" in r.stdout, r.stdout + r.stderr
