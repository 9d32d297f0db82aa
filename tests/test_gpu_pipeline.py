"""End-to-end clustering stage (BASELINE config 5 shape, scaled down): precomputed embeddings -> AHC -> VBx ->
centroids -> constrained assignment, device pipeline vs the CPU restatement of OfflineDiarizerManager.cluster."""
import numpy as np
import pytest
from conftest import speaker_mixture

pytestmark = pytest.mark.gpu


def synth_session(n_chunks, speakers, seed, d=256, dr=128):
    """3 local speaker slots per chunk (OfflineDiarizerTypes.swift:46-55); embeddings = unit speaker centres + noise."""
    rng = np.random.default_rng(seed)
    n = 3 * n_chunks
    centers = rng.standard_normal((speakers, d))
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_chunks)]).reshape(-1)
    emb = (centers[spk] + 0.03 * rng.standard_normal((n, d))).astype(np.float32)
    phi = np.linspace(2.0, 1.0, dr)                                  # between-speaker variances of the PLDA space
    means = rng.standard_normal((speakers, dr)) * np.sqrt(phi)         # VBx's generative model: rho ~ N(m_speaker, I)
    rho = means[spk] + rng.standard_normal((n, dr))
    chunks = np.repeat(np.arange(n_chunks), 3)
    return emb, rho, chunks, phi, spk


@pytest.mark.parametrize("n_chunks,speakers,seed", [(300, 4, 0), (700, 6, 1)])
def test_cluster_stage_matches_cpu_restatement(fa, gpu_ctx, oracle_mod, n_chunks, speakers, seed):
    emb, rho, chunks, phi, spk = synth_session(n_chunks, speakers, seed)
    emb[5] = np.nan                                     # filtered from training (:591-611), still assigned at the end
    res = fa.cluster_embeddings_stagewise(emb, rho, chunks, phi, ctx=gpu_ctx)
    one = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=gpu_ctx)       # the single device-resident call (fa_offline_cluster)
    assert one.assignments == res.assignments and np.array_equal(one.centroids, res.centroids)   # same cores, same bits
    assert one.info["training_rows"] == len(emb) - 1 and one.info["constrained"] == 1 and one.info["was_adjusted"] == 0
    ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi)
    assert res.initial_clusters == ref["initial"].tolist()              # AHC labels bit-exact
    assert one.info["initial_clusters"] == len(set(ref["initial"].tolist()))
    assert res.centroids.shape == ref["centroids"].shape
    np.testing.assert_allclose(res.centroids, ref["centroids"], rtol=0, atol=1e-9)   # VBx gamma differs at 1e-12 (parity unpinned)
    assert res.assignments == ref["assignments"].tolist()
    # the synthetic speakers are recovered (up to relabelling), co-chunk slots are distinct
    lab = np.asarray(res.assignments)
    keep = np.arange(len(lab)) != 5
    assert len(set(zip(spk[keep].tolist(), lab[keep].tolist()))) == speakers
    assert all(len(set(lab[3 * c:3 * c + 3].tolist())) == 3 for c in range(n_chunks) if 5 // 3 != c)


def test_single_call_speaker_count_constraints_and_edge_cases(fa, gpu_ctx, oracle_mod):
    """fa_offline_cluster with numSpeakers forced (K-Means fallback, constrained assignment skipped, :355-358), without PLDA
    features (centroids = per-cluster means of the AHC labels), with one row, and with no finite row at all."""
    emb, rho, chunks, phi, spk = synth_session(200, 5, 3)
    for kw in (dict(num_speakers=3), dict(min_speakers=7), dict(max_speakers=2)):
        cfg = fa.OfflineClusteringConfig(**kw)
        one = fa.cluster_embeddings(emb, rho, chunks, phi, cfg, ctx=gpu_ctx)
        ref = oracle_mod.cluster_embeddings(emb, rho, chunks, phi, **kw)
        st = fa.cluster_embeddings_stagewise(emb, rho, chunks, phi, cfg, ctx=gpu_ctx)
        assert one.info["was_adjusted"] == int(ref["was_adjusted"]) == int(st.vbx.was_adjusted) == 1
        assert one.assignments == st.assignments == ref["assignments"].tolist()
        np.testing.assert_array_equal(one.centroids, st.centroids)
    # no PLDA features: AHC labels -> cluster means -> assignment
    one = fa.cluster_embeddings(emb, np.zeros((len(emb), 0)), chunks, phi, ctx=gpu_ctx)
    init = oracle_mod.ahc_cluster(emb.astype(np.float64), 0.6)
    cen = np.stack([np.cumsum(emb.astype(np.float64)[init == k], axis=0)[-1] / (init == k).sum() for k in sorted(set(init.tolist()))])
    np.testing.assert_array_equal(one.centroids, cen)
    assert one.assignments == oracle_mod.constrained_assign(oracle_mod.centroid_scores(emb.astype(np.float64), cen), chunks).tolist()
    # a single embedding; all rows non-finite (training set falls back to all rows, :606-609)
    one = fa.cluster_embeddings(emb[:1], rho[:1], chunks[:1], phi, ctx=gpu_ctx)
    assert one.assignments == [0] and one.centroids.shape == (1, 256)
    with pytest.raises(ValueError):
        fa.cluster_embeddings(emb[:0], rho[:0], chunks[:0], phi, ctx=gpu_ctx)


def test_batched_recordings_equal_single_calls(fa, gpu_ctx):
    """fa_offline_cluster_batch: several recordings in one call (their merge chains advance together) give, per recording, exactly
    what fa_offline_cluster gives — ragged sizes, a NaN row, a one-row recording; an empty recording fails alone."""
    sessions = [synth_session(260, 4, 0), synth_session(90, 3, 1), synth_session(400, 6, 2), synth_session(33, 3, 3)]
    phi = sessions[0][3]
    sessions[2][0][17] = np.nan
    recs = [(e, r, c) for e, r, c, _, _ in sessions]
    recs.append((sessions[1][0][:1], sessions[1][1][:1], sessions[1][2][:1]))                 # one embedding
    recs.append((sessions[1][0][:0], sessions[1][1][:0], sessions[1][2][:0]))                 # none: noSpeechDetected for that one
    st, out = fa.cluster_embeddings_batch(recs, phi, ctx=gpu_ctx)
    assert st[:-1] == [0] * (len(recs) - 1) and st[-1] == 1 and out[-1] is None
    for (e, r, c), got in zip(recs[:-1], out[:-1]):
        one = fa.cluster_embeddings(e, r, c, phi, ctx=gpu_ctx)
        assert got.assignments == one.assignments
        np.testing.assert_array_equal(got.centroids, one.centroids)
        for key in ("training_rows", "initial_clusters", "vbx_iterations", "was_adjusted", "constrained"):
            assert got.info[key] == one.info[key], key
    assert out[2].info["training_rows"] == len(recs[2][0]) - 1
    # the same recordings RESIDENT on the device (fa_offline_cluster_batch_dev: nothing uploaded): bit for bit the host-pointer call
    import torch
    dev = [(torch.from_numpy(np.ascontiguousarray(e)).cuda(), torch.from_numpy(np.ascontiguousarray(r)).cuda(), c) for e, r, c in recs[:-1]]
    st_d, out_d = fa.cluster_embeddings_batch(dev, phi, ctx=gpu_ctx)
    assert st_d == st[:-1]
    for got, want in zip(out_d, out[:-1]):
        assert got.assignments == want.assignments
        np.testing.assert_array_equal(got.centroids, want.centroids)
        for key in ("training_rows", "initial_clusters", "vbx_iterations", "was_adjusted", "constrained"):
            assert got.info[key] == want.info[key], key
    # forced speaker count goes through the K-Means fallback per recording
    cfg = fa.OfflineClusteringConfig(num_speakers=2)
    st2, out2 = fa.cluster_embeddings_batch(recs[:2], phi, cfg, ctx=gpu_ctx)
    for (e, r, c), got in zip(recs[:2], out2):
        one = fa.cluster_embeddings(e, r, c, phi, cfg, ctx=gpu_ctx)
        assert st2 == [0, 0] and got.assignments == one.assignments and got.info["was_adjusted"] == one.info["was_adjusted"] == 1
