#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" = one pass of the batched STFT->mel featurizer over BASELINE.json configs[1]
(1024 x 15 s 16 kHz chunks per GPU, synthetic PCM already resident in HBM, output [B,128,1501] fp32).
`value` = audio hours featurized per second over all ranks (weak scaling: every rank owns its own 1024 chunks,
no data-path collective).  Rank 0 prints ONE JSON line which also carries
  roofline      — the mel kernel's measured HBM fraction (algorithmic bytes / HIP-event kernel time / 8 TB/s)
  cpu_baseline  — the CPU oracle (a restatement of the Swift/Accelerate path, NOT Apple's vDSP) timed on this box: mel on 1 and
                  8 threads (value = 1 thread), CTC greedy and VBx restatements beside it
  ctc           — greedy CTC decode on [T=1500, V=1024] matrices (BASELINE configs[3]); at N > 1 the 10 000 matrices are SHARDED
                  over the ranks (strong scaling) and the token ids are gathered on rank 0 over RCCL
  ahc_50k       — centroid-linkage AHC on 50 000 x 256 embeddings (the metric's second half; rank 0): the numpy-seeded input whose
                  reference dendrogram digest is committed (tests/golden/ahc_full_iid_50000.json), bit_exact_vs_reference_digest
  ahc_batch     — 16 recordings x 5400 x 256 through fa_ahc_linkage_batch vs sequential calls (rank 0)
  e2e_8h        — BASELINE configs[4]: 8 h audio -> mel -> precomputed embeddings -> AHC + VBx + assignment (fa_offline_cluster);
                  one 8 h recording per rank (replicas: the merge chain of one recording does not shard), labels gathered on rank 0
  featurized_plus_clustered_audio_hours_per_s — the metric's first half on the e2e leg (all ranks)
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK_SAMPLES = 240000          # 15 s @ 16 kHz
CHUNKS_PER_GPU = 1024
MEL_BYTES_PER_CHUNK = 240000 * 4 + 128 * 1501 * 4   # 1 728 512 B
CTC_BYTES_PER_MATRIX = 1500 * 1024 * 4               # 6 144 000 B
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: 8.0 TB/s spec
MEL_KERNEL_SOURCES = ("mel.hip", "mel_v4.inc", "mel_pk.h", "mel_core.h")


def synth_pcm(torch, n_chunks, seed):
    """U(-1,1)*0.1 + two sinusoids, generated on the device."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    n = n_chunks * CHUNK_SAMPLES
    x = (torch.rand(n, generator=g, device="cuda", dtype=torch.float32) * 2 - 1) * 0.1
    t = torch.arange(CHUNK_SAMPLES, device="cuda", dtype=torch.float32) / 16000.0
    tone = 0.3 * torch.sin(2 * np.pi * 440.0 * t) + 0.2 * torch.sin(2 * np.pi * 3000.0 * t)
    x.view(n_chunks, CHUNK_SAMPLES).add_(tone)
    return x


def mel_kernel_sources_sha256():
    h = hashlib.sha256()
    for f in MEL_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "fluidaudio_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def measured_traffic():
    """HBM bytes per launch of the mel kernel from the committed rocprofv3 PMC passes (profiles/*_mel_pmc.json, written by
    scripts/pmc_summary.py + scripts/gpu_mel_pmc.sh: 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction of
    MI355X_MICROARCH.md).  PMC collection needs its own rocprofv3 runs, so the figure is read from the newest summary —
    and REFUSED (None) when the kernel sources have changed since it was measured (kernel_sources_sha256 in the summary)."""
    import glob
    now = mel_kernel_sources_sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mel_pmc.json")), reverse=True):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except Exception:  # noqa: BLE001
            continue
        if j.get("kernel_sources_sha256") == now and j.get("hbm_traffic_bytes_per_launch"):
            return j["hbm_traffic_bytes_per_launch"], {"file": os.path.relpath(f, ROOT), "kernel_sources_sha256": now}
    return None, {"file": None, "note": "no PMC summary for the present kernel sources (sha256 %s...)" % now[:12]}


def _timed_pool(fn, items, threads):
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    if threads == 1:
        for it in items:
            fn(it)
    else:
        with ThreadPoolExecutor(threads) as ex:   # the oracle calls are ctypes calls: the GIL is released inside them
            list(ex.map(fn, items))
    return time.perf_counter() - t0


def cpu_baselines(budget_s=8.0):
    """CPU restatements (oracle/, `kind: port`: Swift/Accelerate cannot run on this box) on bounded samples of the same
    workloads, 1 thread and 8 threads.  The top-level value is the 1-thread mel rate."""
    import oracle
    rng = np.random.default_rng(1234)
    t = np.arange(CHUNK_SAMPLES) / 16000.0
    chunk = (rng.uniform(-1, 1, CHUNK_SAMPLES) * 0.1 + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    oracle.mel_flat(chunk[:16000])
    t1 = _timed_pool(lambda _: oracle.mel_flat(chunk), range(1), 1)
    n1 = max(2, min(256, int(budget_s / max(t1, 1e-3))))
    e1 = _timed_pool(lambda _: oracle.mel_flat(chunk), range(n1), 1)
    n8 = 8 * max(1, n1 // 2)
    e8 = _timed_pool(lambda _: oracle.mel_flat(chunk), range(n8), 8)
    out = {"value": n1 * 15.0 / 3600.0 / e1, "unit": "audio_hours/s", "cores": 1, "kind": "port",
           "sample": f"{n1} x 15 s chunks, oracle/fa_oracle.c computeFlat restatement (fp32 radix-2 FFT + dense 128x257 filterbank), "
                     f"{e1:.1f} s on 1 of {os.cpu_count()} host cores; Swift/Accelerate itself cannot run on this box",
           "mel_8_threads": {"value": n8 * 15.0 / 3600.0 / e8, "unit": "audio_hours/s", "cores": 8, "sample": f"{n8} chunks in {e8:.1f} s"}}
    # CTC greedy (LogitsArgmax.swift:16-55 + CtcDecoder.swift:45-70 restated): [1500, 1024] fp32 matrices
    lg = rng.standard_normal((8, 1500, 1024)).astype(np.float32)
    lg[:, :, 1023] += 2.0
    oracle.ctc_greedy(lg[0], 1023)
    reps = 6
    c1 = _timed_pool(lambda i: oracle.ctc_greedy(lg[i % 8], 1023), range(8 * reps), 1)
    c8 = _timed_pool(lambda i: oracle.ctc_greedy(lg[i % 8], 1023), range(8 * reps * 4), 8)
    out["ctc_greedy"] = {"unit": "matrices/s", "threads_1": 8 * reps / c1, "threads_8": 8 * reps * 4 / c8, "kind": "port",
                         "sample": f"{8 * reps} / {8 * reps * 4} matrices [1500,1024] fp32, oracle argmax + collapse"}
    # VBx (VBxClustering.swift:167-664 restated): N = 6000 frames x 128, 12 initial clusters, <= 20 iterations
    n, spk = 6000, 12
    lab = (np.arange(n) % spk).astype(np.int32)
    phi = np.linspace(2.0, 1.0, 128)
    rho = (rng.standard_normal((spk, 128)) * np.sqrt(phi))[lab] + rng.standard_normal((n, 128))
    v1 = _timed_pool(lambda _: oracle.vbx_refine(rho, lab, phi), range(2), 1)
    v8 = _timed_pool(lambda _: oracle.vbx_refine(rho, lab, phi), range(16), 8)
    out["vbx"] = {"unit": "refinements/s (6000 x 128, 12 clusters)", "threads_1": 2 / v1, "threads_8": 16 / v8, "kind": "port",
                  "sample": "oracle VBx restatement (scalar C, no BLAS); 8 threads = 8 independent recordings"}
    return out


def ahc_leg(fa, ctx, torch, n=50000, d=256, ref_n=3000):
    import ctypes as C
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from ahc_full_inputs import ahc_input, dendrogram_digest, sha256
    out = {}
    gold_path = os.path.join(ROOT, "tests", "golden", f"ahc_full_iid_{n}.json")
    gold = json.load(open(gold_path)) if os.path.exists(gold_path) else None
    xh = ahc_input("iid", n, d)                      # numpy PCG64 seed 0, bytes reproducible on any box
    x = torch.from_numpy(xh).cuda()
    z = torch.zeros((n - 1, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):  # first call allocates the 20 GB workspace; second is the steady-state number
        stats = fa._lib.AhcStats()
        t0 = time.perf_counter()
        st = fa.lib().fa_ahc_linkage(ctx.handle, C.c_void_p(x.data_ptr()), n, d, C.c_void_p(z.data_ptr()), (n - 1) * 4,
                                     fa.AHC_MODE_AUTO, 1, C.byref(stats))
        wall = time.perf_counter() - t0
        out["first_call_s" if rep == 0 else "seconds"] = wall
        if st != 0:
            return {"status": int(st), "error": ctx.last_error()}
    s = stats.as_dict()
    zz = z.cpu().numpy()
    dig = dendrogram_digest(zz)
    out.update({"n": n, "d": d, "distribution": "iid N(0,1) rows, L2-normalised, numpy default_rng(0) (tests/golden/ahc_full_inputs.py)", "inputs": "resident in HBM",
                "device_init_ms": s["init_ms"], "device_merge_ms": s["merge_ms"], "rounds": s["rounds"], "rescans": s["rescans"], "windows": s["windows"],
                "exact_fallback": s["exact_fallback"], "height_inversions": dig["height_inversions"],
                "us_per_round": 1e3 * s["merge_ms"] / max(1, s["rounds"]),
                "dendrogram_sha256": dig["dendrogram_sha256"],
                "bit_exact_vs_reference_digest": None if gold is None else bool(gold["input_sha256"] == sha256(xh) and gold["dendrogram_sha256"] == dig["dendrogram_sha256"]),
                "reference_digest": None if gold is None else {"file": os.path.relpath(gold_path, ROOT), "reference_seconds_1_core": gold["reference_seconds_1_core"]}})
    # the reference's own C++ (oracle/_ref, 1 thread) on a bounded size, next to the GPU at the same size
    xs = xh[:ref_n]
    t0 = time.perf_counter()
    sr, zr = oracle.linkage_ref(xs)
    out["cpu_reference"] = {"n": ref_n, "seconds": time.perf_counter() - t0, "cores": 1, "kind": "reference",
                            "note": "FastClusterWrapper.cpp built -O2 from /root/reference (oracle/_ref); the full 50k x 256 run took "
                                    "937 s on 1 core when the committed digest was generated"}
    t0 = time.perf_counter()
    sg, zg = fa.linkage(xs, ctx=ctx)
    out["gpu_same_n"] = {"n": ref_n, "seconds": time.perf_counter() - t0, "bit_exact_vs_reference": bool(sr == 0 and sg == 0 and np.array_equal(zr, zg))}
    return out


def ahc_batch_leg(fa, ctx, recordings=16, n=5400, d=256, speakers=8):
    """Many medium-sized recordings: fa_ahc_linkage_batch (one launch = one round of every recording) vs sequential calls."""
    import oracle
    rng = np.random.default_rng(21)
    probs = []
    for _ in range(recordings):
        c = rng.standard_normal((speakers, d))
        c /= np.linalg.norm(c, axis=1, keepdims=True)
        x = c[np.arange(n) % speakers] + 0.03 * rng.standard_normal((n, d))
        probs.append(x / np.linalg.norm(x, axis=1, keepdims=True))
    fa.linkage_batch(probs[:2], ctx=ctx)
    t0 = time.perf_counter()
    st, zs, stats = fa.linkage_batch(probs, ctx=ctx, return_stats=True)
    tb = time.perf_counter() - t0
    fa.linkage(probs[0], ctx=ctx)
    t0 = time.perf_counter()
    seq = [fa.linkage(p, ctx=ctx) for p in probs]
    ts = time.perf_counter() - t0
    sr, zr = oracle.linkage_ref(probs[0][:2000])
    sg, zg = fa.linkage_batch([probs[0][:2000], probs[1][:1500]], ctx=ctx)
    return {"recordings": recordings, "embeddings_each": n, "d": d, "batch_s": tb, "batch_device_ms": stats[0]["total_ms"], "sequential_s": ts,
            "speedup_vs_sequential": ts / tb, "statuses_ok": all(s == 0 for s in st),
            "identical_to_sequential": all(np.array_equal(z, s[1]) for z, s in zip(zs, seq)),
            "bit_exact_vs_reference_at_2000": bool(sr == 0 and sg[0] == 0 and np.array_equal(zr, zg[0])),
            "clustered_audio_hours_per_s": recordings * (n / 3 * 2.0 / 3600.0) / tb,
            "note": "host-pointer entries (PCIe copies included); n embeddings = n/3 two-second windows of 3 local speaker slots"}


def ctc_leg(fa, ctx, torch, dist, rank, world, total, steps=3):
    """BASELINE configs[3]: `total` matrices [1500, 1024] fp32 sharded over the ranks (contiguous slices, no data-path
    collective); every rank times its own passes, the slowest rank sets the rate; token ids gather on rank 0 (RCCL)."""
    T, V = 1500, 1024
    lo, hi = fa.shard_range(total, rank, world)
    batch = hi - lo
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    x = torch.randn((batch, T, V), generator=g, device="cuda", dtype=torch.float32)
    x[:, :, V - 1] += 2.0
    tok = torch.zeros((batch, T), dtype=torch.int32, device="cuda")
    lens = torch.zeros(batch, dtype=torch.int32, device="cuda")
    stream = torch.cuda.ExternalStream(ctx.stream)
    torch.cuda.synchronize()
    fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens, order=False)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(steps):
        fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens, order=False)
    e1.record(stream)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = e0.elapsed_time(e1) / steps
    t_gather = None
    gathered = None
    if dist is not None:
        tt = torch.tensor([wall, ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ms = float(tt[0]), float(tt[1])
        rows_n = min(batch, 64)                      # the gather of ALL ids is 4 bytes x 1320 tokens per matrix; a 64-matrix sample per rank shows the path
        tk, ln = tok[:rows_n].cpu().numpy(), lens[:rows_n].cpu().numpy()
        rows = [tk[i, :ln[i]] for i in range(rows_n)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = fa.gather_ragged_int32(rows, dist, dst=0)
        t_gather = time.perf_counter() - t0
        gathered = None if got is None else len(got)
    gbs_rank = batch * CTC_BYTES_PER_MATRIX / (ms * 1e-3) / 1e9
    out = {"matrices": total, "matrices_per_rank": batch, "T": T, "V": V, "dtype": "f32", "ms_per_pass": ms, "wall_ms_per_pass": 1e3 * wall,
           "matrices_per_s": total / wall, "audio_hours_per_s": total * 15.0 / 3600.0 / wall, "scaling": "strong" if world > 1 else "single",
           "roofline": {"bound": "hbm", "achieved": gbs_rank, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_rank / HBM_PEAK_GBS, "traffic": None,
                        "note": "per GPU (slowest rank)"},
           "mean_tokens_per_matrix": float(lens.float().mean()), "gather_token_ids_s": t_gather, "gathered_rows_on_rank0": gathered}
    if world == 1:   # the row kernel next to it (§8f-3): log-softmax with temperature / blank bias, one read + one write of the matrix
        try:
            sub = x[: min(batch, 2500)]
            o = torch.empty_like(sub)
            fa.ctc_log_probs_dev(ctx, sub, 1.0, 0.0, V - 1, d_out=o, order=False)
            ctx.synchronize()
            e0.record(stream)
            for _ in range(steps):
                fa.ctc_log_probs_dev(ctx, sub, 1.0, 0.0, V - 1, d_out=o, order=False)
            e1.record(stream)
            ctx.synchronize()
            ms2 = e0.elapsed_time(e1) / steps
            g2 = 2 * sub.shape[0] * CTC_BYTES_PER_MATRIX / (ms2 * 1e-3) / 1e9
            out["log_softmax"] = {"matrices": int(sub.shape[0]), "ms_per_pass": ms2,
                                  "roofline": {"bound": "hbm", "achieved": g2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g2 / HBM_PEAK_GBS, "traffic": None},
                                  "algorithmic_bytes_per_matrix": 2 * CTC_BYTES_PER_MATRIX}
        except Exception as e:  # noqa: BLE001
            out["log_softmax"] = {"error": repr(e)}
    return out


def e2e_leg(fa, ctx, torch, dist, rank, world, hours=8.0, speakers=12):
    """BASELINE configs[4]: `hours` of synthetic 16 kHz audio per rank -> mel (15 s chunks) -> precomputed embeddings
    (3 local speaker slots per 2 s step, OfflineDiarizerTypes.swift:46-55) -> AHC + VBx + centroids + constrained
    assignment in ONE library call (fa_offline_cluster).  One recording per rank (the merge chain of a recording does not
    shard); the labels gather on rank 0.  Embeddings/PLDA features are synthetic (the reference computes them with CoreML nets)."""
    n_chunks15 = int(hours * 3600 / 15)
    d_pcm = synth_pcm(torch, n_chunks15, 99 + rank)
    mel = fa.AudioMelSpectrogram(ctx=ctx)
    plan = mel.plan(np.arange(n_chunks15 + 1, dtype=np.int64) * CHUNK_SAMPLES, layout="mel_major")
    d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
    d_len = torch.empty(n_chunks15, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    plan.execute(d_pcm, d_out, d_len, order=False)
    ctx.synchronize()
    rng = np.random.default_rng(5 + rank)
    n_win = int(hours * 3600 / 2)
    n = 3 * n_win
    centers = rng.standard_normal((speakers, 256))
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_win)]).reshape(-1)
    emb = (centers[spk] + 0.03 * rng.standard_normal((n, 256))).astype(np.float32)
    phi = np.linspace(2.0, 1.0, 128)
    rho = (rng.standard_normal((speakers, 128)) * np.sqrt(phi))[spk] + rng.standard_normal((n, 128))
    chunks = np.repeat(np.arange(n_win), 3)
    fa.cluster_embeddings(emb, rho, chunks, phi, ctx=ctx)   # warm-up at full size: the context's 15 GB linkage workspace is allocated once (0.4 - 2.5 s of hipMalloc) and kept
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    plan.execute(d_pcm, d_out, d_len, order=False)
    ctx.synchronize()
    t_mel = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=ctx)
    t_cl = time.perf_counter() - t0
    del d_out, d_pcm
    lab = np.asarray(res.assignments)
    pure = len(set(zip(spk.tolist(), lab.tolist()))) == speakers
    t_all = t_mel + t_cl
    gathered = None
    if dist is not None:
        tt = torch.tensor([t_all, t_mel, t_cl], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_all, t_mel, t_cl = float(tt[0]), float(tt[1]), float(tt[2])
        ok = torch.tensor([1.0 if pure else 0.0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        pure = bool(ok.item() > 0.5)
        got = fa.gather_ragged_int32([lab.astype(np.int32)], dist, dst=0)
        gathered = None if got is None else [int(len(g)) for g in got]
    out = {"audio_hours_per_rank": hours, "recordings": world, "mel_chunks_per_rank": n_chunks15, "mel_s": t_mel, "embeddings_per_recording": n, "cluster_s": t_cl,
           "stages_s": res.timings, "speakers_true": speakers, "clusters_found": int(res.centroids.shape[0]),
           "labels_match_speakers": bool(pure), "audio_hours_per_s": world * hours / t_all, "gathered_label_rows": gathered,
           "note": "fa_offline_cluster: embeddings and PLDA features go up once (host pointers, PCIe included), intermediates stay in HBM; mel inputs resident in HBM"}
    if world == 1:
        # the speaker-count fallback on the same embeddings: best-of-10 K-Means to speakers - 2 (VBxClustering.swift:716-722)
        emb64 = emb.astype(np.float64)
        fa.KMeansClustering.cluster_with_centroids_n_init(emb64[:3000], speakers - 2, 100, 10, 0, ctx=ctx)
        t0 = time.perf_counter()
        km, _ = fa.KMeansClustering.cluster_with_centroids_n_init(emb64, speakers - 2, 100, 10, 0, ctx=ctx)
        out["kmeans_fallback_s"] = time.perf_counter() - t0
        out["kmeans_clusters"] = len(set(km))
    return out


def e2e_many_leg(fa, ctx, torch, recordings=16, hours_each=1.0, speakers=8):
    """The serving shape of configs[4]: MANY recordings (a batch job over files) instead of one 8 h recording — mel over all their
    15 s chunks in one launch, then fa_offline_cluster_batch (the merge chains of all recordings advance together)."""
    n_chunks15 = int(recordings * hours_each * 3600 / 15)
    d_pcm = synth_pcm(torch, n_chunks15, 7)
    mel = fa.AudioMelSpectrogram(ctx=ctx)
    plan = mel.plan(np.arange(n_chunks15 + 1, dtype=np.int64) * CHUNK_SAMPLES, layout="mel_major")
    d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
    d_len = torch.empty(n_chunks15, dtype=torch.int32, device="cuda")
    phi = np.linspace(2.0, 1.0, 128)
    recs, truth = [], []
    for r in range(recordings):
        rng = np.random.default_rng(100 + r)
        n_win = int(hours_each * 3600 / 2)
        n = 3 * n_win
        centers = rng.standard_normal((speakers, 256))
        centers /= np.linalg.norm(centers, axis=1, keepdims=True)
        spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_win)]).reshape(-1)
        emb = (centers[spk] + 0.03 * rng.standard_normal((n, 256))).astype(np.float32)
        rho = (rng.standard_normal((speakers, 128)) * np.sqrt(phi))[spk] + rng.standard_normal((n, 128))
        recs.append((emb, rho, np.repeat(np.arange(n_win), 3)))
        truth.append(spk)
    plan.execute(d_pcm, d_out, d_len, order=False)
    ctx.synchronize()
    fa.cluster_embeddings_batch(recs, phi, ctx=ctx)   # warm-up at full size (workspace of all recordings)
    t0 = time.perf_counter()
    plan.execute(d_pcm, d_out, d_len, order=False)
    ctx.synchronize()
    t_mel = time.perf_counter() - t0
    t0 = time.perf_counter()
    st, out = fa.cluster_embeddings_batch(recs, phi, ctx=ctx)
    t_cl = time.perf_counter() - t0
    t0 = time.perf_counter()
    seq = [fa.cluster_embeddings(e, r, c, phi, ctx=ctx) for e, r, c in recs[:4]]
    t_seq4 = time.perf_counter() - t0
    pure = all(s == 0 and len(set(zip(t.tolist(), o.assignments))) == speakers for s, t, o in zip(st, truth, out))
    same = all(a.assignments == b.assignments for a, b in zip(seq, out[:4]))
    hours = recordings * hours_each
    return {"recordings": recordings, "hours_each": hours_each, "embeddings_per_recording": len(truth[0]), "mel_s": t_mel, "cluster_batch_s": t_cl,
            "cluster_sequential_s_extrapolated": t_seq4 * recordings / 4, "labels_match_speakers": bool(pure), "equals_single_calls": bool(same),
            "audio_hours_per_s": hours / (t_mel + t_cl)}


def beam_leg(fa, ctx, torch, batch=512, frames=1500, vocab=1025):
    """CTC prefix beam search + word-level ARPA LM (§8 f3): beam 100, 40 token candidates, one workgroup per utterance."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(batch, frames, vocab, device="cuda", generator=g) * 3.0
    x[:, :, vocab - 1] += 4.0
    lp = torch.log_softmax(x, dim=-1).contiguous()
    del x
    words = ["the", "cat", "sat", "dog", "on", "mat", "a", "in", "of", "to"]
    voc = {v: ("\u2581" + words[v % len(words)] if v % 3 == 0 else "abcdefgh"[v % 8]) for v in range(vocab - 1)}
    arpa = "\\data\\\n\\1-grams:\n" + "".join(f"-{1 + 0.1 * i:.1f}\t{w}\t-0.3\n" for i, w in enumerate(words)) + "\\2-grams:\n" + \
        "".join(f"-0.{5 + i}\t{words[i]}\t{words[(i + 1) % len(words)]}\n" for i in range(len(words))) + "\\end\\\n"
    lm = fa.ARPALanguageModel(arpa, ctx=ctx)
    vocabulary = fa.CtcVocabulary(voc, vocab, ctx)
    tok = torch.zeros(batch, frames, dtype=torch.int32, device="cuda")
    lens = torch.zeros(batch, dtype=torch.int32, device="cuda")
    sc = torch.zeros(batch, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def run():
        ctx.check(fa.lib().fa_ctc_beam_search_batch_dev(ctx.handle, lp.data_ptr(), batch, frames, vocab, vocab, frames * vocab, None, vocabulary.handle,
                                                        lm.handle, 100, 0.3, 0.0, vocab - 1, 40, tok.data_ptr(), lens.data_ptr(), sc.data_ptr()), "beam")
        torch.cuda.synchronize()
    run()
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    return {"workload": f"{batch} x [{frames},{vocab}] log-probs, beam 100, 40 candidates, ARPA LM", "seconds": dt, "utterances_per_s": batch / dt,
            "audio_hours_per_s": batch * frames * 0.01 / 3600 / dt, "us_per_frame_step": dt / frames * 1e6, "mean_tokens": float(lens.float().mean())}


def sharded_start_leg(fa, ctx, torch, dist, rank, world, n=50000, d=256):
    """SURVEY.md §8e, one 50 k problem: all-gather X over RCCL, per-rank row slab of the nearest-neighbour table
    (fa_ahc_row_minima), all-gather of (min, idx) — next to the same table computed by ONE rank.  The merge chain stays on one GPU."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from ahc_full_inputs import ahc_input
    x = ahc_input("iid", n, d)
    d_x = torch.from_numpy(x).cuda()

    def slab_dev(_x_all, lo, hi):
        m = torch.empty(hi - lo, dtype=torch.float64, device="cuda")
        a = torch.empty(hi - lo, dtype=torch.int32, device="cuda")
        ctx.check(fa.lib().fa_ahc_row_minima(ctx.handle, C.c_void_p(d_x.data_ptr()), n, d, lo, hi, C.c_void_p(m.data_ptr()), C.c_void_p(a.data_ptr()), 1), "fa_ahc_row_minima")
        ctx.synchronize()
        return m.cpu().numpy(), a.cpu().numpy()

    lo, hi = fa.shard_range(n, rank, world)
    slab_dev(None, lo, min(hi, lo + 256))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    m, a = fa.sharding.row_minima_sharded(x[lo:hi], lo, n, slab_dev, dist)
    t_sh = time.perf_counter() - t0
    out = {"n": n, "d": d, "ranks": world, "sharded_s": t_sh}
    if rank == 0:
        t0 = time.perf_counter()
        m1, a1 = slab_dev(None, 0, n)
        out["single_rank_s"] = time.perf_counter() - t0
        out["tables_equal"] = bool(np.array_equal(m, m1) and np.array_equal(a, a1))
        out["note"] = ("exact (difference-form) fp64 distances; the production start-up is the Gram-form MFMA kernel (40 ms at 50k on one GPU) and the "
                       "serial merge chain (570 ms) does not shard: see DESIGN.md §4")
    if dist is not None:
        dist.barrier()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clock-warm-s", type=float, default=0.3,
                    help="seconds of the same launches before the W warm-up steps: the device leaves idle clocks only after ~30 ms "
                         "of sustained load (launch time falls from 0.95 to 0.68 ms over the first ~40 launches, scripts/mel_variance.py)")
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_GPU, help="15 s chunks per GPU per step (BASELINE config: 1024)")
    ap.add_argument("--skip-ahc", action="store_true")
    ap.add_argument("--skip-ctc", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-e2e", action="store_true")
    ap.add_argument("--skip-beam", action="store_true")
    ap.add_argument("--skip-sharded-start", action="store_true", help="N > 1 only: the sharded nearest-neighbour start-up of one 50k problem")
    ap.add_argument("--ctc-matrices", type=int, default=10000)
    args = ap.parse_args()

    import torch
    import fluidaudio_amd as fa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    ctx = fa.default_context(local_rank)
    stream = torch.cuda.ExternalStream(ctx.stream)
    B = args.chunks
    d_pcm = synth_pcm(torch, B, 1234 + rank)
    offsets = np.arange(B + 1, dtype=np.int64) * CHUNK_SAMPLES
    mel = fa.AudioMelSpectrogram(ctx=ctx)           # NeMo config: 128 mels, n_fft 512, hop 160, win 400, preemph 0.97
    plan = mel.plan(offsets, layout="mel_major")     # computeFlat layout [B, 128, 1501]
    d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
    d_len = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # inside the timed loops the launches go straight to the context's stream (order=False): inputs are complete (synchronised
    # above) and nothing on torch's stream touches the buffers until the loop has been synchronised
    t_warm = time.perf_counter() + max(0.0, args.clock_warm_s)   # set-up: bring the device to sustained clocks (not timed, not a step)
    while time.perf_counter() < t_warm:
        plan.execute(d_pcm, d_out, d_len, order=False)
        ctx.synchronize()
    for _ in range(args.warmup):
        plan.execute(d_pcm, d_out, d_len, order=False)
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        plan.execute(d_pcm, d_out, d_len, order=False)
        ev[i + 1].record(stream)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
        dist.barrier()
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    kernel_ms_avg = float(np.mean(kernel_ms))
    assert int(d_len[0]) == 1501 and bool(torch.isfinite(d_out[B // 2]).all())

    hours = world * B * 15.0 / 3600.0
    value = hours * args.steps / elapsed
    ach = B * MEL_BYTES_PER_CHUNK / (kernel_ms_avg * 1e-3) / 1e9
    traffic, traffic_source = measured_traffic()
    line = {
        "metric": "audio hours/sec featurized (batched STFT->mel, 1024 x 15 s chunks per GPU); featurized+clustered in "
                  "featurized_plus_clustered_audio_hours_per_s; AHC wall-clock @ 50k x 256 in ahc_50k",
        "value": value, "unit": "audio_hours/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched STFT->mel, 1024 x 15 s 16 kHz chunks per GPU, NeMo config "
                               "(n_fft 512, hop 160, win 400, 128 mels, preemph 0.97), output [B,128,1501] fp32, inputs resident in HBM",
                   "chunks_per_gpu": B, "realtime_factor": value * 3600.0, "clock_warm_s": args.clock_warm_s, "parallelism": f"dp{world} (independent utterance shards, no collective)"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "mel_kernel_v4<MEL_MAJOR>", "kernel_ms_avg": kernel_ms_avg, "kernel_ms_min": float(np.min(kernel_ms)),
                     "algorithmic_bytes_per_launch": B * MEL_BYTES_PER_CHUNK,
                     "note": "the kernel is VALU/LDS-issue bound, not HBM bound: 0.33 ms of pure VALU issue at the measured 4 cycles per wave64 instruction "
                             "(profiles/r02_ubench_peak.txt) vs 0.22 ms of HBM time; see DESIGN.md §3.1"},
    }
    solo = world == 1
    del d_out, d_pcm
    torch.cuda.empty_cache()
    if solo and not args.skip_cpu and rank == 0:
        line["cpu_baseline"] = cpu_baselines()
    if not args.skip_ctc:
        try:
            r = ctc_leg(fa, ctx, torch, dist, rank, world, args.ctc_matrices)
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)}
        line["ctc"] = r
        torch.cuda.empty_cache()
    if not args.skip_e2e:
        try:
            r = e2e_leg(fa, ctx, torch, dist, rank, world)
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)}
        line["e2e_8h"] = r
        line["featurized_plus_clustered_audio_hours_per_s"] = r.get("audio_hours_per_s") if isinstance(r, dict) else None
        torch.cuda.empty_cache()
    if world > 1 and not args.skip_sharded_start:
        try:
            line["ahc_sharded_start"] = sharded_start_leg(fa, ctx, torch, dist, rank, world)
        except Exception as e:  # noqa: BLE001
            line["ahc_sharded_start"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if solo and not args.skip_ahc:
        try:
            line["ahc_50k"] = ahc_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["ahc_50k"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            line["ahc_batch"] = ahc_batch_leg(fa, ctx)
        except Exception as e:  # noqa: BLE001
            line["ahc_batch"] = {"error": repr(e)}
    if solo and not args.skip_e2e:
        torch.cuda.empty_cache()
        try:
            line["e2e_16x1h"] = e2e_many_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["e2e_16x1h"] = {"error": repr(e)}
    if solo and not args.skip_beam:
        torch.cuda.empty_cache()
        try:
            line["beam_search"] = beam_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["beam_search"] = {"error": repr(e)}
    print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
