#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" = one pass of the batched STFT->mel featurizer over BASELINE.json configs[1]
(1024 x 15 s 16 kHz chunks per GPU, synthetic PCM already resident in HBM, output [B,128,1501] fp32).
`value` = audio hours featurized per second over all ranks (weak scaling: every rank owns its own 1024 chunks,
no data-path collective).  Rank 0 prints ONE JSON line which also carries
  roofline      — the mel kernel's measured HBM fraction (algorithmic bytes / HIP-event kernel time / 8 TB/s)
  cpu_baseline  — the CPU oracle (a restatement of the Swift/Accelerate path, NOT Apple's vDSP) timed on this box
  ahc_50k       — wall-clock of centroid-linkage AHC on 50 000 x 256 embeddings (the metric's second half; rank 0)
  ctc           — greedy CTC decode rate on [T=1500, V=1024] matrices (BASELINE configs[3]; rank 0)
  e2e_8h        — BASELINE configs[4] on one GPU: 8 h audio -> mel -> precomputed embeddings -> AHC + VBx + assignment (rank 0)
"""
import argparse
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


def synth_pcm(torch, n_chunks, seed):
    """U(-1,1)*0.1 + two sinusoids, generated on the device."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    n = n_chunks * CHUNK_SAMPLES
    x = (torch.rand(n, generator=g, device="cuda", dtype=torch.float32) * 2 - 1) * 0.1
    t = torch.arange(CHUNK_SAMPLES, device="cuda", dtype=torch.float32) / 16000.0
    tone = 0.3 * torch.sin(2 * np.pi * 440.0 * t) + 0.2 * torch.sin(2 * np.pi * 3000.0 * t)
    x.view(n_chunks, CHUNK_SAMPLES).add_(tone)
    return x


def measured_traffic():
    """HBM bytes per launch of the mel kernel from the committed rocprofv3 PMC passes (profiles/*_mel_pmc.json, written by
    scripts/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction of MI355X_MICROARCH.md); None if absent.
    PMC collection needs its own rocprofv3 runs, so bench.py reports the value measured for the committed kernel."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mel_pmc.json"))):
        try:
            with open(f) as fh:
                best = json.load(fh).get("hbm_traffic_bytes_per_launch", best)
        except Exception:  # noqa: BLE001
            pass
    return best


def cpu_mel_baseline(budget_s=12.0):
    """Time the CPU oracle (1 thread) on a bounded sample of the same workload."""
    import oracle
    rng = np.random.default_rng(1234)
    t = np.arange(CHUNK_SAMPLES) / 16000.0
    chunk = (rng.uniform(-1, 1, CHUNK_SAMPLES) * 0.1 + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    oracle.mel_flat(chunk[:16000])
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.mel_flat(chunk)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 256:
            break
    return {"value": n * 15.0 / 3600.0 / el, "unit": "audio_hours/s", "cores": 1, "kind": "port",
            "sample": f"{n} x 15 s chunks, oracle/fa_oracle.c computeFlat restatement (fp32 radix-2 FFT + dense 128x257 filterbank), "
                      f"{el:.1f} s on 1 of {os.cpu_count()} host cores; Swift/Accelerate itself cannot run on this box"}


def ahc_leg(fa, ctx, torch, n=50000, d=256, ref_n=3000):
    import ctypes as C
    import oracle
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float64)
    x /= x.norm(dim=1, keepdim=True)
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
    out.update({"n": n, "d": d, "distribution": "iid N(0,1) rows, L2-normalised, seed 0", "inputs": "resident in HBM",
                "device_init_ms": s["init_ms"], "device_merge_ms": s["merge_ms"], "rounds": s["rounds"], "rescans": s["rescans"],
                "exact_fallback": s["exact_fallback"], "height_inversions": int((np.diff(zz[:, 2]) < 0).sum()),
                "us_per_round": 1e3 * s["merge_ms"] / max(1, s["rounds"])})
    # the reference's own C++ (oracle/_ref, 1 thread) on a bounded size, next to the GPU at the same size
    xs = x[:ref_n].cpu().numpy()
    t0 = time.perf_counter()
    sr, zr = oracle.linkage_ref(xs)
    out["cpu_reference"] = {"n": ref_n, "seconds": time.perf_counter() - t0, "cores": 1, "kind": "reference",
                            "note": "FastClusterWrapper.cpp built -O2 from /root/reference (oracle/_ref); "
                                    "50k x 256 on 1 core measured at 1001.7 s in BASELINE.md §2"}
    t0 = time.perf_counter()
    sg, zg = fa.linkage(xs, ctx=ctx)
    out["gpu_same_n"] = {"n": ref_n, "seconds": time.perf_counter() - t0, "bit_exact_vs_reference": bool(sr == 0 and sg == 0 and np.array_equal(zr, zg))}
    return out


def ctc_leg(fa, ctx, torch, batch, steps=3):
    T, V = 1500, 1024
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((batch, T, V), generator=g, device="cuda", dtype=torch.float32)
    x[:, :, V - 1] += 2.0
    tok = torch.zeros((batch, T), dtype=torch.int32, device="cuda")
    lens = torch.zeros(batch, dtype=torch.int32, device="cuda")
    stream = torch.cuda.ExternalStream(ctx.stream)
    torch.cuda.synchronize()
    fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens)
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens)
    e1.record(stream)
    ctx.synchronize()
    ms = e0.elapsed_time(e1) / steps
    gbs = batch * CTC_BYTES_PER_MATRIX / (ms * 1e-3) / 1e9
    # the row kernel next to it (§8f-3): log-softmax with temperature / blank bias, one read + one write of the matrix
    lsm = None
    try:
        out = torch.empty_like(x)
        fa.ctc_log_probs_dev(ctx, x, 1.0, 0.0, V - 1, d_out=out)
        ctx.synchronize()
        e0.record(stream)
        for _ in range(steps):
            fa.ctc_log_probs_dev(ctx, x, 1.0, 0.0, V - 1, d_out=out)
        e1.record(stream)
        ctx.synchronize()
        ms2 = e0.elapsed_time(e1) / steps
        g2 = 2 * batch * CTC_BYTES_PER_MATRIX / (ms2 * 1e-3) / 1e9
        lsm = {"ms_per_pass": ms2, "roofline": {"bound": "hbm", "achieved": g2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g2 / HBM_PEAK_GBS,
                                                "traffic": None}, "algorithmic_bytes_per_matrix": 2 * CTC_BYTES_PER_MATRIX}
        del out
    except Exception as e:  # noqa: BLE001
        lsm = {"error": repr(e)}
    return {"log_softmax": lsm, "matrices": batch, "T": T, "V": V, "dtype": "f32", "ms_per_pass": ms, "matrices_per_s": batch / (ms * 1e-3),
            "audio_hours_per_s": batch * 15.0 / 3600.0 / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None},
            "mean_tokens_per_matrix": float(lens.float().mean())}


def e2e_leg(fa, ctx, torch, hours=8.0, speakers=12):
    """BASELINE configs[4] on ONE GPU: `hours` of synthetic 16 kHz audio -> mel (15 s chunks) -> precomputed embeddings
    (3 local speaker slots per 2 s step, OfflineDiarizerTypes.swift:46-55) -> AHC + VBx + centroids + constrained
    assignment.  Embeddings/PLDA features are synthetic (the reference computes them with CoreML nets, out of scope)."""
    n_chunks15 = int(hours * 3600 / 15)
    d_pcm = synth_pcm(torch, n_chunks15, 99)
    mel = fa.AudioMelSpectrogram(ctx=ctx)
    plan = mel.plan(np.arange(n_chunks15 + 1, dtype=np.int64) * CHUNK_SAMPLES, layout="mel_major")
    d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
    d_len = torch.empty(n_chunks15, dtype=torch.int32, device="cuda")
    plan.execute(d_pcm, d_out, d_len)
    ctx.synchronize()
    t0 = time.perf_counter()
    plan.execute(d_pcm, d_out, d_len)
    ctx.synchronize()
    t_mel = time.perf_counter() - t0
    del d_out, d_pcm
    torch.cuda.empty_cache()
    rng = np.random.default_rng(5)
    n_win = int(hours * 3600 / 2)
    n = 3 * n_win
    centers = rng.standard_normal((speakers, 256))
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    spk = np.stack([rng.permutation(speakers)[:3] for _ in range(n_win)]).reshape(-1)
    emb = (centers[spk] + 0.03 * rng.standard_normal((n, 256))).astype(np.float32)
    phi = np.linspace(2.0, 1.0, 128)
    rho = (rng.standard_normal((speakers, 128)) * np.sqrt(phi))[spk] + rng.standard_normal((n, 128))
    chunks = np.repeat(np.arange(n_win), 3)
    fa.cluster_embeddings(emb[:3000], rho[:3000], chunks[:3000], phi, ctx=ctx)   # warm-up (workspace, code objects)
    t0 = time.perf_counter()
    res = fa.cluster_embeddings(emb, rho, chunks, phi, ctx=ctx)
    t_cl = time.perf_counter() - t0
    lab = np.asarray(res.assignments)
    pure = len(set(zip(spk.tolist(), lab.tolist()))) == speakers
    # the speaker-count fallback on the same embeddings: best-of-10 K-Means to speakers - 2 (VBxClustering.swift:716-722)
    emb64 = emb.astype(np.float64)
    fa.KMeansClustering.cluster_with_centroids_n_init(emb64[:3000], speakers - 2, 100, 10, 0, ctx=ctx)
    t0 = time.perf_counter()
    det = {}
    km, _ = fa.KMeansClustering.cluster_with_centroids_n_init(emb64, speakers - 2, 100, 10, 0, ctx=ctx, details=det)
    t_km = time.perf_counter() - t0
    return {"audio_hours": hours, "kmeans_fallback_s": t_km, "kmeans_clusters": len(set(km)), "mel_chunks": n_chunks15, "mel_s": t_mel, "embeddings": n, "cluster_s": t_cl,
            "stages_s": res.timings, "speakers_true": speakers, "clusters_found": int(res.centroids.shape[0]),
            "labels_match_speakers": bool(pure), "audio_hours_per_s": hours / (t_mel + t_cl),
            "note": "host-pointer clustering entries (PCIe copies included); mel inputs resident in HBM"}


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

    t_warm = time.perf_counter() + max(0.0, args.clock_warm_s)   # set-up: bring the device to sustained clocks (not timed, not a step)
    while time.perf_counter() < t_warm:
        plan.execute(d_pcm, d_out, d_len)
        ctx.synchronize()
    for _ in range(args.warmup):
        plan.execute(d_pcm, d_out, d_len)
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        plan.execute(d_pcm, d_out, d_len)
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

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    hours = world * B * 15.0 / 3600.0
    value = hours * args.steps / elapsed
    ach = B * MEL_BYTES_PER_CHUNK / (kernel_ms_avg * 1e-3) / 1e9
    line = {
        "metric": "audio hours/sec featurized (batched STFT->mel, 1024 x 15 s chunks per GPU); AHC wall-clock @ 50k x 256 in ahc_50k",
        "value": value, "unit": "audio_hours/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched STFT->mel, 1024 x 15 s 16 kHz chunks per GPU, NeMo config "
                               "(n_fft 512, hop 160, win 400, 128 mels, preemph 0.97), output [B,128,1501] fp32, inputs resident in HBM",
                   "chunks_per_gpu": B, "realtime_factor": value * 3600.0, "clock_warm_s": args.clock_warm_s, "parallelism": f"dp{world} (independent utterance shards, no collective)"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic(),
                     "kernel": "mel_kernel<MEL_MAJOR>", "kernel_ms_avg": kernel_ms_avg, "kernel_ms_min": float(np.min(kernel_ms)),
                     "algorithmic_bytes_per_launch": B * MEL_BYTES_PER_CHUNK},
    }
    solo = world == 1  # baseline / extra legs only at N=1 (rank 0), so multi-GPU runs stay short
    if solo and not args.skip_cpu:
        line["cpu_baseline"] = cpu_mel_baseline()
    del d_out, d_pcm
    torch.cuda.empty_cache()
    if solo and not args.skip_ctc:
        try:
            line["ctc"] = ctc_leg(fa, ctx, torch, args.ctc_matrices)
        except Exception as e:  # noqa: BLE001
            line["ctc"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if solo and not args.skip_ahc:
        try:
            line["ahc_50k"] = ahc_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["ahc_50k"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if solo and not args.skip_e2e:
        try:
            line["e2e_8h"] = e2e_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["e2e_8h"] = {"error": repr(e)}
    print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
