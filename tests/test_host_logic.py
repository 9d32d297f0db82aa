"""Host-side logic of the library that runs without a GPU: frame arithmetic, tables, dendrogram cut, Python mirrors."""
import ctypes as C

import numpy as np
from conftest import same_partition


def test_frame_arithmetic_matches_oracle(fa, oracle_mod):
    L = fa._lib
    for pm, pre in ((L.MEL_PAD_CENTER, False), (L.MEL_PAD_PREPADDED, True)):
        cfg = L.MelConfig()
        fa.lib().fa_mel_default_config(C.byref(cfg))
        cfg.padding_mode = pm
        for n in (0, 1, 47, 48, 159, 160, 351, 352, 353, 511, 512, 513, 16000, 240000, 12345):
            assert fa.lib().fa_mel_num_frames(C.byref(cfg), n) == oracle_mod.mel_frames(oracle_mod.MelConfig(), n, pre), (pm, n)
    cfg.padding_mode = L.MEL_PAD_LEGACY
    assert fa.lib().fa_mel_num_frames(C.byref(cfg), 16000) == 98          # AudioMelSpectrogramTests.swift:32-45
    assert fa.lib().fa_mel_num_frames(C.byref(cfg), 300) == 1            # Swift truncating division: 1 + (-100)/160
    assert fa.lib().fa_mel_num_frames(C.byref(cfg), 0) == 0
    cfg.pad_to = 16
    assert fa.lib().fa_mel_padded_frames(C.byref(cfg), 101) == 112


def test_tables_bit_identical_to_oracle(fa, oracle_mod):
    L = fa._lib
    for periodic in (0, 1):
        for n_mels in (128, 80):
            cfg = L.MelConfig()
            fa.lib().fa_mel_default_config(C.byref(cfg))
            cfg.window_periodic, cfg.n_mels = periodic, n_mels
            w = np.zeros(400, np.float32)
            fb = np.zeros((n_mels, 257), np.float32)
            assert fa.lib().fa_mel_hann_window(C.byref(cfg), w.ctypes.data) == 0
            assert fa.lib().fa_mel_filterbank(C.byref(cfg), fb.ctypes.data) == 0
            np.testing.assert_array_equal(w, oracle_mod.hann(400, bool(periodic)))
            np.testing.assert_array_equal(fb, oracle_mod.slaney_filterbank(512, n_mels, 16000))


def random_dendrogram(n, rng):
    """Random merge tree in SciPy format with non-monotone heights (inversions are the norm for centroid linkage)."""
    alive = list(range(n))
    z = np.zeros((n - 1, 4))
    size = {i: 1 for i in range(n)}
    for r in range(n - 1):
        i, j = rng.choice(len(alive), 2, replace=False)
        a, b = alive[i], alive[j]
        z[r] = (min(a, b), max(a, b), rng.uniform(0, 2), size[a] + size[b])
        size[n + r] = size[a] + size[b]
        alive = [v for v in alive if v not in (a, b)] + [n + r]
    return z


def test_cut_matches_oracle_on_random_trees(fa, oracle_mod):
    rng = np.random.default_rng(4)
    for n in (2, 3, 7, 64, 500):
        z = random_dendrogram(n, rng)
        for thr in (0.0, 0.3, 0.6, 1.0, 1.9, 2.5, -1.0, float("nan")):
            got = fa.cut(z, n, thr)
            exp = oracle_mod.ahc_cut(z, n, thr)
            np.testing.assert_array_equal(got, exp)
    assert fa.cut(np.zeros((0, 4)), 1, 0.5).tolist() == [0]


def test_python_mirror_guards_need_no_gpu(fa):
    # AHCClustering.cluster guards (:24-31) are decided on the host
    ahc = fa.AHCClustering.__new__(fa.AHCClustering)
    ahc._ctx, ahc.mode = None, 0
    assert ahc.cluster([], 0.7) == []
    assert ahc.cluster([[], [], []], 0.7) == [0, 0, 0]
    assert ahc.cluster([[1.0, 0.0, 0.0]], 0.7) == [0]
    assert fa.decode_ctc_token_ids([0, 1, 2], {0: "he", 1: "llo", 2: "▁world"}) == "hello world"  # CtcDecoderTests.swift:55-59
    assert fa.ctc_greedy_decode([], {0: "▁hello"}, 1) == ""


def test_shard_ranges_cover_and_balance(fa):
    for n in (0, 1, 7, 8, 1024, 10000):
        for w in (1, 2, 3, 8):
            spans = [fa.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    offs = np.array([0, 10, 25, 25, 40, 100])
    first, local, lo, hi = fa.shard_offsets(offs, 1, 2)
    assert first == 3 and local.tolist() == [0, 15, 75] and (lo, hi) == (25, 100)


def test_speaker_constraints_and_rng_host_functions(fa, oracle_mod):
    """SpeakerCountConstraints.resolve and the seeded draws are pure host functions of the library (no GPU)."""
    from test_oracle_kmeans import test_speaker_constraints_resolve
    cases = test_speaker_constraints_resolve.pytestmark[0].args[1]
    for args, expect in cases:
        c = fa.SpeakerCountConstraints.resolve(*args)
        assert (c.num_speakers, c.min_speakers, c.max_speakers) == expect
    c = fa.SpeakerCountConstraints.resolve(100, None, 5, 10)       # SpeakerCountConstraintsTests.swift:104-136
    assert c.needs_adjustment(3) and c.target_count(3) == 5
    c = fa.SpeakerCountConstraints.resolve(100, None, 2, 5)
    assert c.needs_adjustment(8) and c.target_count(8) == 5 and not c.needs_adjustment(3) and c.target_count(3) == 3
    a, b = fa.SeededRNG(1234), oracle_mod.SeededRNG(1234)
    for bound in (1, 2, 3, 10, 43200, 2 ** 31 + 11, 2 ** 63 + 5, 2 ** 64 - 1):
        for _ in range(50):
            assert a.next_below(bound) == b.next_upper_bound(bound)
    assert a.next() == b.next()


def test_vbx_output_cluster_counts(fa):
    """VBxConstraintTests.swift:53-160: pi census vs clusters that actually win an argmax."""
    V = fa.VBxOutput
    out = V(np.zeros((0, 0)), np.array([0.5, 0.3, 0.2, 1e-9, 1e-10]), [], [], 5, [])
    assert out.active_cluster_count == 3
    assert V(np.zeros((0, 0)), np.zeros(0), [], [], 4, []).active_cluster_count == 4
    g = np.array([[0.7, 0.1, 0.1, 0.05, 0.05], [0.1, 0.7, 0.1, 0.05, 0.05], [0.1, 0.1, 0.7, 0.05, 0.05],
                  [0.6, 0.2, 0.1, 0.05, 0.05], [0.2, 0.6, 0.1, 0.05, 0.05], [0.1, 0.2, 0.6, 0.05, 0.05]])
    out = V(g, np.array([0.4, 0.3, 0.28, 0.01, 0.01]), [], [], 5, [])
    cons = fa.SpeakerCountConstraints.resolve(6, 5)
    assert out.active_cluster_count == 5 and out.assigned_cluster_count == 3
    assert not cons.needs_adjustment(out.active_cluster_count) and cons.needs_adjustment(out.assigned_cluster_count)


def test_htk_filterbank_and_config_validation(fa):
    """fa_mel_filterbank with FA_MEL_SCALE_HTK_NONORM = LuxTtsMelExtractor.htkMelFilterbank (LuxTtsMelExtractor.swift:160-189);
    the extended fa_mel_config fields are validated host-side (no GPU needed: plan creation fails before any device call)."""
    import ctypes as C
    L = fa._lib
    cfg = L.MelConfig(sample_rate=24000, n_mels=100, n_fft=1024, hop=256, win=1024, mel_scale=L.MEL_SCALE_HTK_NONORM, power=1.0)
    got = np.zeros((100, 513), np.float32)
    assert fa.lib().fa_mel_filterbank(C.byref(cfg), got.ctypes.data) == 0
    h2m = lambda hz: 2595 * np.log10(1 + hz / 700)  # noqa: E731
    m2h = lambda m: 700 * (10 ** (m / 2595) - 1)  # noqa: E731
    pts = m2h(h2m(0) + np.arange(102) * (h2m(12000.0) - h2m(0)) / 101)
    fr = np.arange(513) * 12000.0 / 512
    ref = np.stack([np.maximum(0, np.minimum((fr - pts[m]) / (pts[m + 1] - pts[m]), (pts[m + 2] - fr) / (pts[m + 2] - pts[m + 1]))) for m in range(100)])
    np.testing.assert_allclose(got, ref.astype(np.float32), rtol=0, atol=1e-7)
    assert (got >= 0).all() and got.max() <= 1.0 and (got.sum(axis=1) > 0).all()
    # a caller-supplied table wins over mel_scale
    custom = np.ascontiguousarray(np.random.default_rng(0).random((100, 513)), np.float32)
    cfg.filterbank = custom.ctypes.data
    assert fa.lib().fa_mel_filterbank(C.byref(cfg), got.ctypes.data) == 0
    np.testing.assert_array_equal(got, custom)
    # frame counts of the extension follow the same centre formula: win == n_fft -> 1 + n / hop (torch.stft, LuxTts :68)
    cfg.filterbank = None
    for n in (1, 255, 256, 103936):
        assert fa.lib().fa_mel_num_frames(C.byref(cfg), n) == 1 + n // 256


def test_vbx_shard_geometry_needs_no_gpu(fa):
    """fa_vbx_shard_range / _chunk_doubles / _slices (host arithmetic of the sharded VBx protocol) and their Python mirror: the ranks of a
    world cover [0, T) without gaps in rank order, own whole slices of ceil(T / 64) frames, and a world size that does not divide 64 owns
    nothing (create refuses it)."""
    from fluidaudio_amd import _lib as L
    from fluidaudio_amd.sharding import VBX_SLICES, vbx_shard_frames
    lib = L.lib()
    assert lib.fa_vbx_shard_slices() == VBX_SLICES == 64
    for T in (1, 63, 64, 65, 1000, 43200, 345600):
        per = -(-T // 64)
        for world in (1, 2, 4, 8, 16, 32, 64):
            end = 0
            for rank in range(world):
                lo, hi = C.c_int64(), C.c_int64()
                lib.fa_vbx_shard_range(T, rank, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == vbx_shard_frames(T, rank, world)
                assert lo.value == end and (lo.value % per == 0 or lo.value == T)
                end = hi.value
            assert end == T
    lo, hi = C.c_int64(7), C.c_int64(7)
    lib.fa_vbx_shard_range(1000, 0, 3, C.byref(lo), C.byref(hi))
    assert (lo.value, hi.value) == (0, 0)
    assert lib.fa_vbx_shard_chunk_doubles(24, 128, 8) == 8 * (24 * 129 + 1)
    assert lib.fa_vbx_shard_chunk_doubles(24, 128, 3) == 0
    assert lib.fa_vbx_shard_chunk_doubles(24, 128, 1) * 8 == 64 * (24 * 129 + 1) * 8   # the all-gather of one iteration, in bytes


def test_default_configs_equal_the_reference_defaults(fa):
    """The *_default_config entries are pure host functions; their values are the reference's defaults: AudioMelSpectrogram.init
    (Shared/AudioMelSpectrogram.swift:59-70), TdtConfig (TdtConfig.swift:13-26) and OfflineDiarizerConfig.default as its own test pins it
    (OfflineModuleTests.swift:10-21: threshold 0.6, Fa 0.07, Fb 0.8, 20 iterations)."""
    L = fa._lib
    m = L.MelConfig()
    fa.lib().fa_mel_default_config(C.byref(m))
    assert (m.sample_rate, m.n_mels, m.n_fft, m.hop, m.win, m.pad_to) == (16000, 128, 512, 160, 400, 0)
    assert m.preemph == np.float32(0.97) and m.log_floor == np.float32(2.0 ** -24)
    assert (m.floor_mode, m.window_periodic, m.padding_mode, m.layout) == (L.MEL_FLOOR_ADDITIVE, 0, L.MEL_PAD_CENTER, 0)
    t = L.TdtConfig()
    fa.lib().fa_tdt_default_config(C.byref(t))
    assert (t.blank_id, t.max_symbols_per_step, t.max_tokens_per_chunk, t.consecutive_blank_limit) == (8192, 10, 150, 5)
    assert t.n_duration_bins == 5 and list(t.duration_bins)[:5] == [0, 1, 2, 3, 4]
    o = L.OfflineClusterConfig()
    fa.lib().fa_offline_cluster_default_config(C.byref(o))
    assert (o.clustering_threshold, o.warm_start_fa, o.warm_start_fb, o.max_vbx_iterations) == (0.6, 0.07, 0.8, 20)
    assert o.convergence_tolerance == 1e-4 and o.constrained_assignment == 1                       # VBxClustering.swift:653-659
    assert (o.num_speakers, o.min_speakers, o.max_speakers) == (-1, -1, -1)                           # nil: no speaker-count constraint


def test_clustering_config_guards_of_the_reference(fa):
    """OfflineDiarizerConfig.validate (OfflineDiarizerTypes.swift:357-408), the clustering / VBx part, as its tests exercise it
    (OfflineModuleTests.swift:10-33: the default passes, threshold 2.5 throws naming clustering.threshold); the guards fire before
    any device is touched, so this runs without a GPU."""
    import pytest
    fa.OfflineClusteringConfig().validate()
    x = np.zeros((4, 256), np.float32)
    for kw, word in ((dict(clustering_threshold=2.5), "clustering.threshold"), (dict(clustering_threshold=0.0), "clustering.threshold"),
                     (dict(clustering_threshold=float("nan")), "clustering.threshold"), (dict(warm_start_fa=0.0), "Fa/Fb"),
                     (dict(warm_start_fb=-1.0), "Fa/Fb"), (dict(max_vbx_iterations=0), "maxVBxIterations"),
                     (dict(convergence_tolerance=0.0), "convergenceTolerance")):
        cfg = fa.OfflineClusteringConfig(**kw)
        for call in (lambda: fa.cluster_embeddings(x, np.zeros((4, 128)), np.zeros(4, np.int32), np.ones(128), cfg),
                     lambda: fa.cluster_embeddings_stagewise(x, np.zeros((4, 128)), np.zeros(4, np.int32), np.ones(128), cfg),
                     lambda: fa.cluster_embeddings_batch([(x, np.zeros((4, 128)), np.zeros(4, np.int32))], np.ones(128), cfg)):
            with pytest.raises(ValueError, match="invalidConfiguration") as e:
                call()
            assert word in str(e.value)
    fa.OfflineClusteringConfig(clustering_threshold=2.0).validate()          # the closed end of (0, 2]


def test_tables_bit_identical_to_oracle_over_the_configurations_callers_use(fa, oracle_mod):
    """Hann window and Slaney bank for the other front ends of the reference (LS-EEND: nFFT = nextPow2(winLength) at 8 kHz,
    LSEENDTypes.swift:55-57; 80-mel Parakeet variants; 24 kHz TTS front ends): fp32 formulas of AudioMelSpectrogram.swift:553-642."""
    L = fa._lib
    for sr, n_fft, win, n_mels in ((16000, 512, 400, 128), (16000, 512, 400, 80), (8000, 256, 200, 23), (8000, 256, 256, 40), (16000, 1024, 1024, 64),
                                   (24000, 1024, 1024, 100), (22050, 2048, 1024, 128), (16000, 64, 64, 8), (48000, 2048, 2048, 256)):
        for periodic in (0, 1):
            cfg = L.MelConfig()
            fa.lib().fa_mel_default_config(C.byref(cfg))
            cfg.sample_rate, cfg.n_fft, cfg.win, cfg.n_mels, cfg.window_periodic = sr, n_fft, win, n_mels, periodic
            w = np.zeros(win, np.float32)
            fb = np.zeros((n_mels, n_fft // 2 + 1), np.float32)
            assert fa.lib().fa_mel_hann_window(C.byref(cfg), w.ctypes.data) == 0
            assert fa.lib().fa_mel_filterbank(C.byref(cfg), fb.ctypes.data) == 0
            np.testing.assert_array_equal(w, oracle_mod.hann(win, bool(periodic)))
            np.testing.assert_array_equal(fb, oracle_mod.slaney_filterbank(n_fft, n_mels, sr))
            assert fb.min() >= 0 and (fb.sum(axis=1) > 0).sum() >= n_mels - 2            # AudioMelSpectrogramTests.swift:93-105


def test_cut_refuses_dendrograms_it_cannot_walk(fa):
    """fa.cut validates before it calls the C entry: a child index at or beyond its own row's node is an out-of-bounds read or a cycle for
    the walk of AHCClustering.swift:124-197 (which only ever sees its own wrapper's output)."""
    import pytest
    rng = np.random.default_rng(1)
    good = random_dendrogram(9, rng)
    fa.check_dendrogram(good, 9)
    for mutate in (lambda z: z.__setitem__((3, 0), 9 + 3),          # its own node: a cycle
                   lambda z: z.__setitem__((2, 1), 9 + 5),          # a later node
                   lambda z: z.__setitem__((0, 0), -1),
                   lambda z: z.__setitem__((4, 1), 1e12),
                   lambda z: z.__setitem__((1, 0), np.nan),
                   lambda z: z.__setitem__((5, 0), z[5, 0] + 0.5),
                   lambda z: z.__setitem__((6, 0), z[6, 1]),        # a node merged with itself
                   lambda z: z.__setitem__((7, 0), z[0, 0])):       # a node merged twice
        z = good.copy()
        mutate(z)
        with pytest.raises(ValueError):
            fa.cut(z, 9, 0.5)
    with pytest.raises(ValueError):
        fa.cut(good[:-1], 9, 0.5)
    assert fa.cut(good, 9, 5.0).tolist() == [0] * 9


def test_c_entry_fa_ahc_cut_refuses_dendrograms_it_cannot_walk(fa):
    """The C entry itself (what a Swift / C host binds, not the Python guard): every damaged matrix comes back INVALID_ARGUMENT without
    an out-of-bounds access or an endless walk, valid ones keep the labels of the walk in AHCClustering.swift:124-197.  Includes the
    shapes the round-3 fuzz run found (a child at or beyond its own row's node, cycles through later rows)."""
    L = fa._lib
    rng = np.random.default_rng(7)

    def raw_cut(z, n, thr=0.5):
        z = np.ascontiguousarray(z, np.float64)
        labels = np.full(max(n, 1), -7, np.int32)
        return fa.lib().fa_ahc_cut(z.ctypes.data, n, float(thr), labels.ctypes.data), labels

    for n in (2, 3, 9, 64, 257):
        good = random_dendrogram(n, rng)
        st, lab = raw_cut(good, n)
        assert st == L.SUCCESS and lab.min() >= 0
        # the labels of the guarded Python path are the labels of the raw entry
        assert lab[:n].tolist() == fa.cut(good, n, 0.5).tolist()
        for trial in range(60):
            z = good.copy()
            r = int(rng.integers(0, n - 1))
            c = int(rng.integers(0, 2))
            kind = trial % 8
            if kind == 0:
                z[r, c] = n + r                                   # its own node
            elif kind == 1:
                z[r, c] = n + r + int(rng.integers(0, n))         # a later (or non-existent) node
            elif kind == 2:
                z[r, c] = -1 - int(rng.integers(0, 5))
            elif kind == 3:
                z[r, c] = [np.nan, np.inf, -np.inf, 1e300][trial // 8 % 4]
            elif kind == 4:
                z[r, c] += 0.25
            elif kind == 5:
                z[r, 0] = z[r, 1]                                 # a node merged with itself
            elif kind == 6:
                if n == 2:
                    continue
                other = (r + 1 + int(rng.integers(0, n - 2))) % (n - 1)
                z[r, c] = z[other, int(rng.integers(0, 2))]       # a node merged twice
                if z[r, 0] == z[r, 1] or z[r, c] >= n + r:
                    pass                                          # still damaged, another way
            else:
                z[r, c] = 2.0 ** 53 + 2 * trial                   # whole number far beyond any node
            st, lab = raw_cut(z, n)
            assert st == L.INVALID_ARGUMENT, (n, trial, kind, st)
    # a two-row cycle: row 0 names node n+1, row 1 names node n (the endless walk of the fuzz run)
    z = np.array([[0, 5, 0.1, 2], [1, 4, 0.2, 3], [2, 3, 0.3, 4]], np.float64)
    assert raw_cut(z, 4)[0] == L.INVALID_ARGUMENT


def test_bench_line_contract_on_the_committed_line():
    """The driver parses ONE JSON line of bench.py: the committed line of the round (profiles/r04_bench_v17.json, written on an MI355X)
    carries every field of the contract with the right types, and bench.py still spells each of them."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r04_bench_v17.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    need = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float, "higher_is_better": bool,
            "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict, "cpu_baseline": dict}
    for k, t in need.items():
        assert isinstance(line[k], t), (k, type(line[k]))
    assert "vs_baseline" in line and line["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic" and "workload" in line["config"]
    assert abs(line["value"] - line["n_gpus"] * line["config"]["hours_per_step_per_gpu"] * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert line["e2e_equals_reference_digest"] is True
    # round 4: the second half of the metric and the other north-star kernels ride in `config` (the driver keeps it whole), every one with its check
    cfg = line["config"]
    assert 0 < cfg["ahc_50k_seconds"] < 1.0 and cfg["ahc_50k_bit_exact_vs_reference_digest"] is True and cfg["ctc_ids_exact"] is True
    for k in ("mel_roofline_frac", "ctc_roofline_frac", "ctc_fp16_roofline_frac", "resample_44k1_roofline_frac", "tdt_roofline_frac"):
        assert 0 < cfg[k] < 1, k
    assert line["tdt"]["ids_equal_cpu_restatement_all_chunks"] is True and line["tdt"]["ids_equal_table_walk_all_chunks"] is True
    assert all(v["within_2e-5"] for v in line["resample"].values() if isinstance(v, dict))
    assert all(v["equal_single_calls"] and v["recording_0_equals_reference_digest"] for k, v in line["e2e_8h_batch"].items() if k.startswith("x"))
    assert line["e2e_8h_hard"]["equals_reference_digest"] is True
    src = open(os.path.join(root, "bench.py")).read()
    for k in list(need) + ["vs_baseline", "bound", "achieved", "peak", "frac", "traffic", "cores", "kind", "sample"]:
        assert f'"{k}"' in src, k
