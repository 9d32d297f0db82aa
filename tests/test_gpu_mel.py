"""GPU parity: HIP mel (through the C ABI) vs the CPU oracle / committed golden vectors.

Tolerance (BASELINE.json north_star: "mel frames within 1e-4 rel-fp32"): |gpu - ref| <= 1e-4 * max(1, |ref|) on the
log-mel values.  Both sides are fp32 with different DFT factorizations; measured max deviation is printed."""
import os

import numpy as np
import pytest
from conftest import synth_audio

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "mel_golden.npz")


def close(got, ref, what=""):
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= 1e-4, f"{what}: max rel err {err.max():.3e}"
    return err.max()


def test_committed_golden_vectors(fa, gpu_ctx):
    g = np.load(GOLD)
    mel = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    m1, l1, n1 = mel.compute_flat(g["a1"])
    assert (l1, n1) == (int(g["len1"]), g["flat1"].shape[1])
    close(m1.reshape(128, n1), g["flat1"], "computeFlat")
    m2, l2, n2 = mel.compute_flat_transposed(g["a2"], last_audio_sample=0.25)
    assert l2 == int(g["len2"])
    close(m2.reshape(n2, 128), g["tr2"], "computeFlatTransposed")
    m3, l3, n3 = mel.compute_flat_transposed(g["a2"], padding_mode="prePadded")
    assert l3 == int(g["len3"])
    close(m3.reshape(n3, 128), g["pre3"], "prePadded")


@pytest.mark.parametrize("n", [1, 47, 48, 160, 399, 400, 511, 513, 2560, 16000, 24001, 160000])
def test_single_utterance_vs_oracle(fa, gpu_ctx, oracle_mod, n):
    a = synth_audio(n, seed=n)
    mel = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    got, ml, nf = mel.compute_flat(a, last_audio_sample=0.3)
    ref, rml, rnf = oracle_mod.mel_flat(a, last=0.3)
    assert (ml, nf) == (rml, rnf)
    close(got.reshape(128, nf), ref, f"flat n={n}")
    got, ml, nf = mel.compute_flat_transposed(a)
    ref, rml, rnf = oracle_mod.mel_flat_transposed(a)
    assert (ml, nf) == (rml, rnf)
    close(got.reshape(nf, 128), ref, f"transposed n={n}")


def test_guard_and_padding(fa, gpu_ctx, oracle_mod):
    mel = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    m, ml, nf = mel.compute_flat(np.zeros(0, np.float32))
    assert ml == 0 and nf == 1 and m.size == 128 and not m.any()  # :199-201
    mel16 = fa.AudioMelSpectrogram(pad_to=16, ctx=gpu_ctx)
    a = synth_audio(16000, 3)
    got, ml, nf = mel16.compute_flat(a)
    ref, rml, rnf = oracle_mod.mel_flat(a, oracle_mod.MelConfig(pad_to=16))
    assert (ml, nf) == (rml, rnf) == (101, 112)
    got = got.reshape(128, nf)
    close(got, ref, "padTo")
    assert not got[:, 101:].any()
    # silence -> log floor, all negative (AudioMelSpectrogramTests.swift:104-122)
    s, ml, nf = mel.compute_flat(np.zeros(16000, np.float32))
    np.testing.assert_allclose(s.reshape(128, nf)[:, :ml], np.log(np.float32(2.0 ** -24)), rtol=1e-6)


def test_other_configs(fa, gpu_ctx, oracle_mod):
    a = synth_audio(20000, 8)
    # LS-EEND flavour (LSEENDPreprocessor.swift:70-82): preemph 0, clamped floor 1e-10, periodic window
    m = fa.AudioMelSpectrogram(preemph=0.0, log_floor=1e-10, log_floor_mode="clamped", window_periodic=True, ctx=gpu_ctx)
    got, ml, nf = m.compute_flat_transposed(a)
    ref, _, _ = oracle_mod.mel_flat_transposed(a, oracle_mod.MelConfig(preemph=0.0, log_floor=1e-10, floor_clamped=True, window_periodic=True))
    close(got.reshape(nf, 128), ref, "ls-eend")
    # 80 mels, hop 128
    m = fa.AudioMelSpectrogram(n_mels=80, hop_length=128, ctx=gpu_ctx)
    got, ml, nf = m.compute_flat(a)
    ref, rml, _ = oracle_mod.mel_flat(a, oracle_mod.MelConfig(n_mels=80, hop=128))
    assert ml == rml
    close(got.reshape(80, nf), ref, "80 mels")
    # expectedFrameCount override + truncated windows (:347, :412)
    got, ml, nf = fa.AudioMelSpectrogram(ctx=gpu_ctx).compute_flat_transposed(a, expected_frame_count=140)
    ref, rml, rnf = oracle_mod.mel_flat_transposed(a, expected_frames=140)
    assert (ml, nf) == (rml, rnf) == (140, 140)
    close(got.reshape(nf, 128), ref, "expected frames")
    # legacy compute(): 1 s -> 98 frames (AudioMelSpectrogramTests.swift:32-45)
    got, T = fa.AudioMelSpectrogram(ctx=gpu_ctx).compute(a[:16000])
    ref, rT = oracle_mod.mel_legacy(a[:16000])
    assert T == rT == 98
    close(got[0], ref, "legacy")


def test_stream_equals_batch(fa, gpu_ctx):
    """SortformerStreamingMelTests.swift:100-132 on the device path: chunked .prePadded frames with carried
    lastAudioSample equal the one-shot .center frames within 1e-5."""
    a = synth_audio(16000 * 3, 21)
    mel = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    batch, T, _ = mel.compute_flat_transposed(a)
    batch = batch.reshape(-1, 128)
    padded = np.concatenate([np.zeros(256, np.float32), a, np.zeros(256, np.float32)])
    # stream in 10-frame hops: each call sees 512 + 9*160 samples; pre-emphasis state = sample before the chunk
    frames, pos, t = [], 0, 0
    while t < T:
        k = min(10, T - t)
        seg = padded[pos:pos + 512 + (k - 1) * 160]
        last = padded[pos - 1] if pos > 0 else 0.0
        # inside the left pad the reference's stream is zeros, so pre-emphasis of the first real sample uses 0
        out, ml, nf = mel.compute_flat_transposed(seg, last_audio_sample=float(last), padding_mode="prePadded")
        assert ml == k
        frames.append(out.reshape(nf, 128)[:k])
        pos += k * 160
        t += k
    stream = np.concatenate(frames)
    # frames that reach into the RIGHT pad differ by construction (the stream pre-emphasises the first pad sample
    # against the last real sample, the batch path pads after pre-emphasis), so the last 4 frames are excluded
    np.testing.assert_allclose(stream[:T - 4], batch[:T - 4], atol=1e-5)


def test_batched_ragged_device_plan(fa, gpu_ctx, oracle_mod):
    import torch
    lens = [16000, 12370, 1, 0, 240000, 4801]
    audios = [synth_audio(n, 100 + i) for i, n in enumerate(lens)]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    lasts = np.linspace(-0.2, 0.2, len(lens)).astype(np.float32)
    mel = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    for layout in ("mel_major", "frame_major"):
        plan = mel.plan(offs, layout=layout)
        d_pcm = torch.from_numpy(np.concatenate(audios)).cuda()
        d_last = torch.from_numpy(lasts).cuda()
        d_out = torch.full(plan.out_shape(), 7.0, dtype=torch.float32, device="cuda")
        d_len = torch.zeros(len(lens), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        plan.execute(d_pcm, d_out, d_len, d_last)
        gpu_ctx.synchronize()
        out, ln = d_out.cpu().numpy(), d_len.cpu().numpy()
        for b, a in enumerate(audios):
            if layout == "mel_major":
                ref, rl, _ = oracle_mod.mel_flat(a, last=float(lasts[b]))
                got = out[b][:, :rl]
                pad = out[b][:, rl:]
                ref = ref[:, :rl]
            else:
                ref, rl, _ = oracle_mod.mel_flat_transposed(a, last=float(lasts[b]))
                got = out[b][:rl]
                pad = out[b][rl:]
                ref = ref[:rl]
            assert ln[b] == rl
            if rl:
                close(got, ref, f"{layout} utt {b}")
            assert not pad.any()  # padValue 0 beyond T (:39)
        plan.close()


def test_full_size_config2_properties(fa, gpu_ctx, oracle_mod):
    """BASELINE config 2 shape (1024 x 15 s) checked through size-independent properties: utterances are independent
    (a batch of identical chunks gives identical rows; one spot-checked against the oracle) and a gain of g shifts
    power-dominated log-mel values by 2 ln g."""
    import torch
    B, n = 1024, 240000
    base = synth_audio(n, 77)
    d_pcm = torch.from_numpy(base).cuda().repeat(B)
    d_pcm[n * 5:n * 6] *= 2.0  # utterance 5 has gain 2
    offs = (np.arange(B + 1, dtype=np.int64) * n)
    mel = fa.AudioMelSpectrogram(ctx=gpu_ctx)
    plan = mel.plan(offs)
    d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    plan.execute(d_pcm, d_out)
    gpu_ctx.synchronize()
    assert plan.total_frames == B * 1501
    out = d_out
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[1023]) and torch.equal(out[3], out[777])
    ref, rl, _ = oracle_mod.mel_flat(base)
    close(out[0].cpu().numpy(), ref, "config-2 row")
    shift = (out[5] - out[0]).cpu().numpy()
    strong = ref > -8.0  # bins where the 2^-24 floor is negligible
    np.testing.assert_allclose(shift[strong], 2 * np.log(2.0), atol=2e-3)
    plan.close()


def test_unified_extractor_per_feature_normalisation(fa, gpu_ctx, oracle_mod):
    """UnifiedMelExtractor.features (UnifiedMelExtractor.swift:52-113): NeMo per_feature normalisation over the valid
    frames, pad frames zero, [1, n_mels, T] layout; batched form against per-window oracle results."""
    import torch
    ws = 16000 * 2
    ex = fa.UnifiedMelExtractor(ws, ctx=gpu_ctx)
    assert ex.total_frames == ws // 160 + 1
    wins, valids = [], [ws, 12345, 160, 100, 0]
    for i, v in enumerate(valids):
        w = np.zeros(ws, np.float32)
        w[:v] = synth_audio(v, 40 + i) if v else 0
        wins.append(w)
    d_mel, vf = ex.features_batch(torch.from_numpy(np.stack(wins)).cuda(), valids)
    got = d_mel.cpu().numpy()
    for i, v in enumerate(valids):
        ref, rv = oracle_mod.unified_mel_features(wins[i], v)
        assert vf[i] == rv == min(v // 160, ex.total_frames)
        assert np.all(got[i][:, rv:] == 0)
        # normalised values are O(1); the oracle sums sequentially in fp32, the kernel as a tree
        np.testing.assert_allclose(got[i], ref, rtol=0, atol=2e-3 if rv > 1 else 1e-6)
        if rv > 8:
            assert np.abs(got[i][:, :rv].mean(1)).max() < 1e-4
    one, v1 = ex.features(wins[1], valids[1])
    assert one.shape == (1, 128, ex.total_frames) and v1 == vf[1]
    np.testing.assert_array_equal(one[0], got[1])
