#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

BASELINE.json metric: "audio hours/sec featurized+clustered per node; AHC wall-clock @ 50k x 256 embeds".
A "step" = BASELINE configs[4] on one GPU: ONE 8 h recording featurized and clustered — batched STFT->mel over its 1 920 x 15 s
chunks (synthetic PCM resident in HBM) + the clustering stage on its 43 200 precomputed embeddings (256-d fp32 + 128-d PLDA
features, resident in HBM): AHC (threshold 0.6) -> VBx -> gamma-weighted centroids -> per-chunk constrained assignment in one
library call (fa_offline_cluster).  `value` = audio hours featurized + clustered per second over all ranks (weak scaling: every
rank owns its own recording; the merge chain of one recording does not shard — DESIGN.md §4).  The session of rank 0 is the one
whose CPU-side results are committed (tests/golden/e2e_8h.json: AHC on the REFERENCE's own linkage build, the rest from the C
restatements); `e2e_equals_reference_digest` says the timed calls reproduced them.

Output (rank 0): one `{"leg": name, "result": {...}}` line per leg as soon as the leg is done, everything again in bench_legs.json,
one `{"summary": {...}}` line, and LAST the result line the driver parses — the contract's keys, scalars only, < 4 KB — with
  roofline      — the dominant kernel of the step (ahc_round_t, one launch per merge): algorithmic bytes per launch / average
                  launch period measured with HIP events around the merge phase, against 8 TB/s; traffic from the committed PMC pass
  cpu_baseline  — the same path on this box's host cores: the reference's linkage build (oracle/_ref) + the C restatements, 1 thread
Legs:
  e2e_8h        — the timed region itself (stages, digest checks, start-up TFLOP/s)
  mel           — BASELINE configs[1]: 1024 x 15 s chunks per GPU, its own HBM roofline (HIP events per launch)
  mel_single_10s— configs[0] shape: one 10 s utterance through the host-pointer entry, p50 / p99 latency
  ctc           — configs[3]: greedy CTC on 10 000 x [1500, 1024] matrices (sharded over the ranks at N > 1), ids verified in-bench
  ahc_50k       — configs[2]: the metric's second half, dendrogram SHA-256 against the reference build's committed digest
  ahc_batch, ahc_ties, e2e_16x1h, e2e_8h_batch, e2e_8h_hard, beam_search — serving-shaped legs (rank 0, N = 1)
  resample, tdt, ctc_fp16 — the other north-star kernels, each with its own roofline and an in-bench check
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

# Hardware queues of the HIP runtime for this process (read when the runtime starts).  Only the optional round-3 leg (--in-flight: four independent chains
# of dependent launches) wants more than the default four: with the streams the other legs created before it, two chains shared a queue (58 instead of
# 87 audio-hours/s).  The serving path of round 4 (uniform batches, at most two streams) runs on the runtime's defaults: nothing is set for it.
if "--in-flight" in sys.argv:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def ahc_ties_leg(fa, ctx, kinds=("dup30", "silence5", "grid64")):
    """What an input with exact ties costs, and that the tie route returns what the REFERENCE BUILD returns: the 8 h session's rows with 30 % duplicated /
    5 % one identical row / on a 1/64 grid (tests/golden/ahc_full_inputs.ahc_tied_input).  AUTO halts at the first tied minimum and the problem runs in
    the reference's selection order (fastcluster_internal.hpp:1685-1799) through the matrix filter; every dendrogram is compared with the SHA-256 of what
    oracle/_ref produced on the same bytes (tests/golden/ahc_tied_<kind>_43200.json, make_ahc_full_digest.py --tied) and timed next to the tie-free rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from ahc_full_inputs import ahc_tied_input, dendrogram_digest, sha256

    def timed(data, reps=2):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            st, z, stats = fa.linkage(data, mode=fa.AHC_MODE_AUTO, ctx=ctx, return_stats=True)
            dt = time.perf_counter() - t0
            if st != 0:
                raise RuntimeError(f"linkage status {st}: {ctx.last_error()}")
            if best is None or dt < best[0]:
                best = (dt, z, stats)
        return best
    x = ahc_tied_input("tie_free")
    n = len(x)
    t_free, _, s_free = timed(x)
    out = {"n": n, "d": x.shape[1], "tie_free_seconds": t_free, "tie_free_reference_order": s_free["reference_order"],
           "note": "host-pointer entries: the upload / download of 88 MB is inside every wall time"}
    for kind in kinds:
        with open(os.path.join(ROOT, "tests", "golden", f"ahc_tied_{kind}_{n}.json")) as f:
            want = json.load(f)
        xd = ahc_tied_input(kind)
        same_input = sha256(xd) == want["input_sha256"]
        t_tie, z_tie, s_tie = timed(xd)
        out[kind] = {"seconds": t_tie, "over_tie_free": t_tie / t_free, "reference_order": s_tie["reference_order"], "us_per_row": 1e3 * s_tie["merge_ms"] / (n - 1),
                     "startup_ms": s_tie["init_ms"], "rescans": s_tie.get("rescans"), "input_regenerated_bit_for_bit": same_input,
                     "equals_reference_digest": bool(same_input and dendrogram_digest(z_tie)["dendrogram_sha256"] == want["dendrogram_sha256"]),
                     "reference_seconds_1_core_when_generated": want["reference_seconds_1_core"]}
    out["tied_seconds"] = out[kinds[0]]["seconds"]                                  # the 30 % duplicates: the figure rounds 4 and 5 quoted
    out["tied_over_tie_free"] = out[kinds[0]]["over_tie_free"]
    out["equals_reference_digest"] = all(out[k]["equals_reference_digest"] for k in kinds)
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


def ctc_leg(fa, ctx, torch, dist, rank, world, total, steps=3, dtype="f32", vocab=1024):
    """BASELINE configs[3]: `total` matrices [1500, 1024] fp32 sharded over the ranks (contiguous slices, no data-path
    collective); every rank times its own passes, the slowest rank sets the rate; token ids gather on rank 0 (RCCL).
    vocab = 1025 is the shape Parakeet CTC really emits (1 024 tokens + blank): rows whose alignment rotates with the frame index."""
    T, V = 1500, vocab
    lo, hi = fa.shard_range(total, rank, world)
    batch = hi - lo
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    x = torch.randn((batch, T, V), generator=g, device="cuda", dtype=torch.float32)
    x[:, :, V - 1] += 2.0
    half = dtype == "f16"
    if half:                                       # LogitsArgmax.swift:31-55 widens fp16 logits to fp32 before the argmax: half the bytes per matrix
        x = x.half()
    esize = 2 if half else 4
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
    bytes_per_matrix = T * V * esize
    gbs_rank = batch * bytes_per_matrix / (ms * 1e-3) / 1e9
    # ---- verification of what was timed: EVERY frame id against torch.argmax on the device, the collapse of 8 matrices against the oracle
    fid = torch.empty((batch, T), dtype=torch.int32, device="cuda")
    fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens, d_frame_ids=fid, order=False)
    ctx.synchronize()
    ids_exact = True
    for b0 in range(0, batch, 1000):
        ids_exact = ids_exact and bool(torch.equal(fid[b0:b0 + 1000].long(), torch.argmax(x[b0:b0 + 1000].float(), dim=-1)))
    rows_exact = None
    if rank == 0:
        import oracle
        rows_exact = True
        for b in list(range(4)) + [batch // 2, batch - 3, batch - 2, batch - 1]:
            ref = oracle.ctc_greedy(x[b].float().cpu().numpy(), V - 1)
            got = tok[b, :int(lens[b])].cpu().numpy()
            rows_exact = rows_exact and bool(np.array_equal(got, ref))
    del fid
    traffic, tsrc = measured_traffic_of("*_ctc_pmc.json", CTC_SOURCES)
    if half or V != 1024:
        traffic, tsrc = None, {"file": None, "note": "the PMC pass ran the fp32 launch of [1500, 1024] matrices"}
    if traffic is not None:
        traffic = traffic * batch / 10000.0          # the PMC pass runs the 10 000-matrix launch; per launch of this rank's share
    out = {"matrices": total, "matrices_per_rank": batch, "T": T, "V": V, "dtype": dtype, "ms_per_pass": ms, "wall_ms_per_pass": 1e3 * wall,
           "matrices_per_s": total / wall, "audio_hours_per_s": total * 15.0 / 3600.0 / wall, "scaling": "strong" if world > 1 else "single",
           "ids_exact": bool(ids_exact), "ids_checked": f"all {batch} x {T} frame ids == torch.argmax on the device", "collapsed_rows_equal_oracle": rows_exact,
           "roofline": {"bound": "hbm", "achieved": gbs_rank, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_rank / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                        "algorithmic_bytes_per_launch": batch * bytes_per_matrix, "note": "per GPU (slowest rank)"},
           "mean_tokens_per_matrix": float(lens.float().mean()), "gather_token_ids_s": t_gather, "gathered_rows_on_rank0": gathered}
    if world == 1 and not half and V == 1024:   # the row kernel next to it (§8f-3): log-softmax with temperature / blank bias, one read + one write of the matrix
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


def e2e_in_flight_leg(fa, torch, in_flight=4, steps=3, hours=8.0):
    """Throughput of ONE GPU with several recordings of configs[4] in flight: `in_flight` host threads, each with its own context (stream,
    workspace) and its own resident inputs, run the clustering stage of an 8 h recording concurrently — the merge chain of one recording keeps
    169 of the 256 CUs busy at one wavefront per SIMD and waits on latency most of the time, so independent chains overlap almost freely
    (profiles/r03_e2e_in_flight.json: 1 / 2 / 3 / 4 / 8 in flight = 32 / 57 / 76 / 87 / 88 audio-hours/s).  Every result is checked
    against the committed digest.  (The headline stays ONE recording per step: that is the latency of a recording.)"""
    import threading
    from e2e_inputs import e2e_session, sha256
    with open(os.path.join(ROOT, "tests", "golden", "e2e_8h.json")) as f:
        gold = json.load(f)
    s = e2e_session(hours, gold["speakers"])
    ctxs = [fa.Context(torch.cuda.current_device()) for _ in range(in_flight)]
    emb = torch.from_numpy(np.ascontiguousarray(s["emb"], np.float32)).cuda()
    rho = torch.from_numpy(np.ascontiguousarray(s["rho"], np.float64)).cuda()
    ok = [True] * in_flight

    def work(k, n):
        for _ in range(n):
            res = fa.cluster_embeddings(emb, rho, s["chunks"], s["phi"], ctx=ctxs[k])
            if hours == gold["hours"] and sha256(np.asarray(res.assignments, np.int32)) != gold["assignments_sha256"]:
                ok[k] = False
    for k in range(in_flight):
        work(k, 1)                      # warm-up: the workspaces (15 GB each)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, steps)) for k in range(in_flight)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    for c in ctxs:
        c.close()                       # releases the workspaces
    return {"recordings_in_flight": in_flight, "steps_each": steps, "hours_each": hours, "wall_s": wall, "audio_hours_per_s": in_flight * steps * hours / wall,
            "s_per_recording": wall / steps, "all_equal_reference_digest": all(ok),
            "note": "clustering stage only (mel of the same audio adds 1.2 ms per recording); inputs resident in HBM; one process, one GPU"}


def e2e_batch_leg(fa, ctx, ks=(2, 4, 8, 12), hours=8.0):
    """Throughput of ONE GPU, queue-independent form: K recordings of configs[4] through ONE fa_offline_cluster_batch call — their merge chains
    advance by ONE launch per round (uniform workspace layout, ahc_round_uni: the problem is the workgroup id in y), on one stream, whatever
    hardware queues the process's other streams occupy (the in-flight leg below depends on them).  Recording k = the session of seed 5 + k;
    recording 0 is digest-checked, every recording is compared with its own single call.  Inputs RESIDENT in HBM like the headline's
    (fa_offline_cluster_batch_dev, round 6); `host_pointers` repeats the largest two sizes through the host-pointer entry, whose time includes the PCIe
    upload of the embeddings (88 MB per recording) — what rounds 4 - 5 reported for this leg."""
    import torch
    from e2e_inputs import e2e_session, sha256
    with open(os.path.join(ROOT, "tests", "golden", "e2e_8h.json")) as f:
        gold = json.load(f)
    phi = None
    recs, singles = [], []
    for k in range(max(ks)):
        s = e2e_session(hours, gold["speakers"], seed=5 + k)
        phi = s["phi"]
        recs.append((s["emb"], s["rho"], s["chunks"]))
        singles.append(np.asarray(fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=ctx).assignments, np.int32))
    out = {"hours_each": hours, "embeddings_each": len(recs[0][0]), "inputs": "resident in HBM (fa_offline_cluster_batch_dev)"}
    dev = [(torch.from_numpy(e).cuda(), torch.from_numpy(r).cuda(), c) for e, r, c in recs]
    torch.cuda.synchronize()
    for k in ks:
        fa.cluster_embeddings_batch(dev[:k], phi, ctx=ctx)                  # warm-up at this size: the K workspaces are one allocation
        t0 = time.perf_counter()
        st, res = fa.cluster_embeddings_batch(dev[:k], phi, ctx=ctx)
        wall = time.perf_counter() - t0
        same = all(s_ == 0 and np.array_equal(np.asarray(r.assignments, np.int32), singles[i]) for i, (s_, r) in enumerate(zip(st, res)))
        a = res[0].info["ahc"]
        out[f"x{k}"] = {"recordings": k, "wall_s": wall, "audio_hours_per_s": k * hours / wall, "equal_single_calls": bool(same),
                        "recording_0_equals_reference_digest": bool(hours == gold["hours"] and sha256(np.asarray(res[0].assignments, np.int32)) == gold["assignments_sha256"]),
                        "us_per_round": 1e3 * a["merge_ms"] / max(1, a["rounds"]), "rounds": a["rounds"], "ahc_init_ms": a["init_ms"], "ahc_merge_ms": a["merge_ms"],
                        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                     "achieved": k * 3 * ((len(recs[0][0]) + 255) // 256 * 256) * 8 / (1e-6 * 1e3 * a["merge_ms"] / max(1, a["rounds"])) / 1e9,
                                     "kernel": "ahc_round_uni (one slot per thread)" if (k if k < 6 else (k + 1) // 2) * ((len(recs[0][0]) + 255) // 256) < 450 else
                                               "ahc_round_uni_c2 (two slots per thread; six or more recordings: two batches side by side)",
                                     "note": "K x (two operand rows + one written row) per launch / launch period (of the caller's batch when two run side by side)"}}
        out[f"x{k}"]["roofline"]["frac"] = out[f"x{k}"]["roofline"]["achieved"] / HBM_PEAK_GBS
    out["host_pointers"] = {}
    for k in sorted(ks)[-2:]:
        t0 = time.perf_counter()
        st, res = fa.cluster_embeddings_batch(recs[:k], phi, ctx=ctx)
        wall = time.perf_counter() - t0
        same = all(s_ == 0 and np.array_equal(np.asarray(r.assignments, np.int32), singles[i]) for i, (s_, r) in enumerate(zip(st, res)))
        out["host_pointers"][f"x{k}"] = {"wall_s": wall, "audio_hours_per_s": k * hours / wall, "equal_single_calls": bool(same)}
    del dev
    ctx.trim()
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
    n_seq = min(4, recordings)
    seq = [fa.cluster_embeddings(e, r, c, phi, ctx=ctx) for e, r, c in recs[:n_seq]]
    t_seq4 = (time.perf_counter() - t0) * 4 / n_seq
    pure = all(s == 0 and len(set(zip(t.tolist(), o.assignments))) == speakers for s, t, o in zip(st, truth, out))
    same = all(a.assignments == b.assignments for a, b in zip(seq, out[:n_seq]))
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

    import ctypes as C
    plan = (C.c_int64 * 4)()
    fa.lib().fa_ctc_beam_plan(batch, frames, vocab, 100, vocab - 1, 40, plan)
    fa.lib().fa_ctx_set_timing(ctx.handle, 1)

    def run():
        t0 = time.perf_counter()
        ctx.check(fa.lib().fa_ctc_beam_search_batch_dev(ctx.handle, lp.data_ptr(), batch, frames, vocab, vocab, frames * vocab, None, vocabulary.handle,
                                                        lm.handle, 100, 0.3, 0.0, vocab - 1, 40, tok.data_ptr(), lens.data_ptr(), sc.data_ptr()), "beam")
        wall = time.perf_counter() - t0                                   # the entry returns after hipStreamSynchronize
        return wall, fa.lib().fa_ctx_last_device_ms(ctx.handle) * 1e-3    # device time: events on ctx.stream behind the call's allocations

    # The walk is a dependent chain per utterance: its time scales with the shader clock, and a device that has idled through the set-up above
    # (3 GB of randn + log_softmax on the host's schedule) starts at idle clocks.  Warm up until ~0.3 s of the same work has run, then repeat.
    cached0 = ctx.workspace_bytes()
    warm = [run()]
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.3 or len(warm) < 3:
        warm.append(run())
    cached1 = ctx.workspace_bytes()
    sclk_before = ctx.sclk_mhz()
    reps = [run() for _ in range(7)]
    sclk_after = ctx.sclk_mhz()
    fa.lib().fa_ctx_set_timing(ctx.handle, 0)
    walls = sorted(r[0] for r in reps)
    devs = sorted(r[1] for r in reps)
    dt = walls[len(walls) // 2]
    return {"workload": f"{batch} x [{frames},{vocab}] log-probs, beam 100, 40 candidates, ARPA LM",
            "seconds": dt, "seconds_min_median_max": [walls[0], dt, walls[-1]], "device_seconds_min_median_max": [devs[0], devs[len(devs) // 2], devs[-1]],
            "timing": "7 repeats after >= 0.3 s of the same call; seconds = median wall-clock of the synchronous entry (allocations included), device_seconds = "
                      "HIP events on the context's stream around the launches of a call (behind its allocations)",
            "first_call_seconds": warm[0][0], "warmup_calls": len(warm),
            "kernel": f"ctc_topk_kernel + ctc_beam_kernel<{plan[3]}, false>", "trie_slots_per_utterance": plan[0], "utterances_per_launch": plan[1], "launches": plan[2],
            "arena_bytes": plan[0] * 8 * plan[1], "context_cache_bytes_before_after_warmup": [cached0, cached1],
            "sclk_mhz_before_after": [sclk_before, sclk_after],
            "utterances_per_s": batch / dt, "audio_hours_per_s": batch * frames * 0.01 / 3600 / dt, "us_per_frame_step": dt / frames * 1e6,
            "device_us_per_frame_step": devs[len(devs) // 2] / frames * 1e6, "mean_tokens": float(lens.float().mean())}


def resample_leg(fa, ctx, torch, seconds=3600):
    """The polyphase resampler (north star; the default path of AudioConverter.swift:60-71,299-370 is Apple's closed AVAudioConverter, so the
    kernel is a labelled extension with scipy.signal.resample_poly as its CPU second opinion): `seconds` of mono fp32 audio resident in HBM
    at 48 / 44.1 / 22.05 / 8 / 96 / 88.2 kHz -> 16 kHz, taps of the pair cached in the context (designed and uploaded by the first call, not per call),
    HIP events on the context's stream.  Algorithmic bytes = 4 (n_in + n_out): every sample read once, every output written once."""
    import ctypes as C
    from scipy import signal
    stream = torch.cuda.ExternalStream(ctx.stream)
    out = {}
    for name, rate, up, down in (("48000->16000", 48000, 1, 3), ("44100->16000", 44100, 160, 441), ("22050->16000", 22050, 320, 441), ("8000->16000", 8000, 2, 1),
                                 ("96000->16000", 96000, 1, 6), ("88200->16000", 88200, 80, 441)):
        n = rate * seconds
        g = torch.Generator(device="cuda").manual_seed(rate)
        x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32) * 0.1
        n_out = int(fa.lib().fa_resample_poly_frames(n, up, down))
        y = torch.empty(n_out, device="cuda", dtype=torch.float32)
        got = C.c_int64()

        def run():
            ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "fa_resample_poly_dev")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(); ctx.synchronize()
        first_ms = 1e3 * (time.perf_counter() - t0)                  # includes the tap design + upload of the pair
        for _ in range(2):
            run()
        ctx.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            run()
        e1.record(stream)
        ctx.synchronize()
        ms = e0.elapsed_time(e1) / reps
        # second opinion on a slice: the first 2 s of output from the first 3 s of input (the FIR reaches 10 max(up, down) input samples)
        k_in, k_out = 3 * rate, 2 * 16000
        ref = signal.resample_poly(x[:k_in].cpu().numpy().astype(np.float64), up, down, window=("kaiser", 5.0))[:k_out]
        err = float(np.max(np.abs(y[:k_out].cpu().numpy() - ref)))
        gbs = 4.0 * (n + n_out) / (ms * 1e-3) / 1e9
        traffic, tsrc = measured_traffic_of(f"*_resample_{rate}_pmc.json", RESAMPLE_SOURCES) if seconds == 3600 else (None, {"file": None, "note": "the PMC passes ran one hour of audio"})
        out[name] = {"up": up, "down": down, "samples_in": n, "samples_out": n_out, "ms_per_pass": ms, "first_call_ms": first_ms,
                     "audio_hours_per_s": seconds / 3600.0 / (ms * 1e-3), "max_abs_err_vs_scipy_first_2s": err, "within_2e-5": bool(err <= 2e-5),
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                                  "algorithmic_bytes_per_launch": 4 * (n + n_out)}}
        del x, y
    out["note"] = (f"{seconds} s of audio per pass, inputs resident in HBM; kernels: register-tiled decimation (48 k), row-tiled polyphase (44.1 k / 22.05 k), "
                   "LDS-staged (8 k); parity unpinned (the reference delegates to AVAudioConverter), scipy.signal.resample_poly is the second opinion")
    return out


def tdt_leg(fa, ctx, torch, B=1024, U=64, T=188, V1=1025, nd=5, dtype="float32"):
    """The TDT greedy walk (TdtDecoderV3.swift:230-467) on joint LOGITS resident in HBM: B chunks of 15 s (T = 188 encoder frames), the joint
    evaluated on a (u, t) grid of U x T cells of W = V1 + nd logits each (the networks themselves are not in the reference tree: synthetic
    logits, ~75 % blanks).  The walk visits ~T + tokens cells per chunk and reads only those rows: algorithmic bytes = visited cells x W x 4.
    Checks in-bench: the walk of ALL chunks against the same walk on decision tables built by torch (argmax / softmax over the whole grid)
    and against the CPU restatement (oracle.tdt_greedy) on those tables."""
    import oracle
    W = V1 + nd
    g = torch.Generator(device="cuda").manual_seed(17)
    SL = 256                                                      # chunks per slice of the generation / verification (the fp32 grid of 4 096 chunks would be 203 GB)
    lg = torch.empty((B, U, T, W), device="cuda", dtype=torch.float16 if dtype == "float16" else torch.float32)
    for b0 in range(0, B, SL):
        part = torch.randn((min(SL, B - b0), U, T, W), generator=g, device="cuda", dtype=torch.float32)
        part[..., V1 - 1] += 4.0
        lg[b0:b0 + SL] = part.to(lg.dtype)
        del part
    enc = np.full(B, T, np.int32)
    stream = torch.cuda.ExternalStream(ctx.stream)
    from fluidaudio_amd.tdt import TdtConfig
    tcfg = TdtConfig(blank_id=V1 - 1)
    res = fa.tdt_decode_logits(lg, V1, enc, config=tcfg, max_out=U, ctx=ctx)
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    from fluidaudio_amd import _lib as L
    cfg = tcfg.c()
    v_enc = torch.from_numpy(enc).cuda()
    o = [torch.zeros((B, U), dtype=torch.int32, device="cuda") for _ in range(3)]
    o_conf = torch.zeros((B, U), dtype=torch.float32, device="cuda")
    o1 = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(4)]
    pp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    dt = L.DTYPE_F16 if dtype == "float16" else L.DTYPE_F32

    def run():
        ctx.check(fa.lib().fa_tdt_greedy_logits_dev(ctx.handle, C.byref(cfg), pp(lg), dt, B, U, T, V1, W, pp(v_enc), None, None, None, None, None, U,
                                                    pp(o[0]), pp(o[1]), pp(o[2]), pp(o_conf), pp(o1[0]), pp(o1[1]), pp(o1[2]), pp(o1[3])), "fa_tdt_greedy_logits_dev")
    torch.cuda.synchronize()
    run(); ctx.synchronize()
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    ctx.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tokens = int(sum(r["count"] for r in res))
    elem = 2 if dtype == "float16" else 4
    # tables for the whole grid on the device (torch, a slice of chunks at a time), walked by the table kernel: same tokens / timestamps / durations for
    # ALL chunks; every chunk against the CPU restatement on the same tables; the restatement also counts its joint evaluations = the rows of W logits
    # the device walk read (one row per decision): the algorithmic bytes of the launch
    same, ok_cpu, visited, longest = True, True, 0, 0
    for b0 in range(0, B, SL):
        x32 = lg[b0:b0 + SL].float()
        tok_t = torch.argmax(x32[..., :V1], dim=-1).to(torch.int32)
        bin_t = torch.argmax(x32[..., V1:], dim=-1).to(torch.int32)
        prob_t = torch.softmax(x32[..., :V1], dim=-1).amax(dim=-1)
        del x32
        tab = fa.tdt_decode_tables(tok_t, bin_t, prob_t, enc[b0:b0 + SL], config=tcfg, max_out=U, ctx=ctx)
        same = same and all(np.array_equal(a["tokens"], b["tokens"]) and np.array_equal(a["timestamps"], b["timestamps"]) and np.array_equal(a["durations"], b["durations"])
                            and a["final_time"] == b["final_time"] for a, b in zip(res[b0:b0 + SL], tab))
        tok_h, bin_h, prob_h = tok_t.cpu().numpy(), bin_t.cpu().numpy(), prob_t.cpu().numpy()
        for i in range(tok_h.shape[0]):
            b = b0 + i
            ref = oracle.tdt_greedy(tok_h[i], bin_h[i], prob_h[i], int(enc[b]), int(enc[b]), 0, False, 0, None, blank_id=V1 - 1, max_out=U)
            visited += ref["joint_calls"]
            longest = max(longest, ref["joint_calls"])
            ok_cpu = ok_cpu and ref["status"] == res[b]["status"] and np.array_equal(ref["tokens"], res[b]["tokens"]) and np.array_equal(ref["timestamps"], res[b]["timestamps"]) \
                and np.array_equal(ref["durations"], res[b]["durations"])
        del tok_t, bin_t, prob_t
    bytes_read = visited * W * elem
    gbs = bytes_read / (ms * 1e-3) / 1e9
    return {"workload": f"{B} chunks x joint logits [U={U}, T={T}, W={W}] {dtype}, greedy TDT walk, logits resident in HBM ({lg.numel() * elem / 1e9:.1f} GB)",
            "ms_per_pass": ms, "chunks_per_s": B / (ms * 1e-3), "audio_hours_per_s": B * 15.0 / 3600.0 / (ms * 1e-3), "tokens_emitted": tokens,
            "ids_equal_table_walk_all_chunks": bool(same), "ids_equal_cpu_restatement_all_chunks": bool(ok_cpu), "rows_read": visited,
            "rows_of_the_longest_chunk": longest, "rows_per_chunk_mean": visited / B,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": measured_traffic_of("*_tdt_pmc.json", TDT_SOURCES)[0] if (B, U, T, V1, dtype) == (1024, 64, 188, 1025, "float32") else None,
                         "algorithmic_bytes_per_launch": bytes_read,
                         "note": "latency-bound by construction: a chunk's walk is ~T + tokens DEPENDENT row reads (argmax of a row decides the next row); "
                                 "the batch of chunks is the parallel axis"}}


AHC_SOURCES = ("ahc_round_body.h", "ahc_ws.h", "ahc_rounds.hip")   # the round kernel of the headline: body, shared definitions, entry kernels
CTC_SOURCES = ("ctc.hip",)
RESAMPLE_SOURCES = ("resample.hip", "resample_geom.h")
TDT_SOURCES = ("tdt.hip",)


def sources_sha256(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "fluidaudio_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def measured_traffic_of(pattern, files):
    """HBM bytes per launch from the newest committed PMC summary matching profiles/<pattern>, refused when the kernel sources changed."""
    import glob
    now = sources_sha256(files)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except Exception:  # noqa: BLE001
            continue
        if j.get("kernel_sources_sha256") == now and j.get("hbm_traffic_bytes_per_launch"):
            return j["hbm_traffic_bytes_per_launch"], {"file": os.path.relpath(f, ROOT), "kernel_sources_sha256": now}
    return None, {"file": None, "note": "no PMC summary for the present kernel sources (sha256 %s...)" % now[:12]}


def e2e_session_for(rank, hours):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from e2e_inputs import e2e_session, input_digest
    s = e2e_session(hours, 12, seed=5 + rank)
    gold = aux = None
    gp = os.path.join(ROOT, "tests", "golden", "e2e_8h.json")
    if rank == 0 and hours == 8.0 and os.path.exists(gp):
        with open(gp) as f:
            gold = json.load(f)
        if gold["input_sha256"] != input_digest(s):
            gold = None                                  # another numpy: the digests do not apply
        else:
            aux = np.load(gp[:-5] + ".npz")
    return s, gold, aux


def mel_single_leg(fa, ctx, calls=200):
    """BASELINE configs[0] shape (the reference's callers are streaming: one utterance per call, StreamingEouAsrManager.swift:558):
    one 10 s / 160 000-sample utterance through the HOST-pointer entry (upload, kernel, download, synchronise), latency per call."""
    import oracle
    rng = np.random.default_rng(11)
    t = np.arange(160000) / 16000.0
    a = (rng.uniform(-1, 1, 160000) * 0.1 + 0.3 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    mel = fa.AudioMelSpectrogram(ctx=ctx)
    for _ in range(10):
        got, ml, nf = mel.compute_flat(a)
    lat = []
    for _ in range(calls):
        t0 = time.perf_counter()
        got, ml, nf = mel.compute_flat(a)
        lat.append(time.perf_counter() - t0)
    lat = np.sort(np.asarray(lat))
    t0 = time.perf_counter()
    ref, rml, rnf = oracle.mel_flat(a)
    t_cpu = time.perf_counter() - t0
    err = float(np.max(np.abs(got.reshape(128, nf) - ref) / np.maximum(1.0, np.abs(ref))))
    return {"samples": 160000, "frames": int(ml), "calls": calls, "p50_ms": 1e3 * float(lat[len(lat) // 2]), "p99_ms": 1e3 * float(lat[int(len(lat) * 0.99) - 1]),
            "min_ms": 1e3 * float(lat[0]), "realtime_factor_p50": 10.0 / float(lat[len(lat) // 2]), "cpu_oracle_1_core_ms": 1e3 * t_cpu,
            "max_rel_err_vs_oracle": err, "within_1e-4": bool(err <= 1e-4 and (ml, nf) == (rml, rnf)),
            "note": "host-pointer entry fa_mel_batch incl. PCIe both ways and the Python ctypes call"}


def vbx_sharded_leg(fa, ctx, torch, dist, rank, world, hours=64.0):
    """the VBx iteration loop sharded over the frame axis (VBxClustering.swift:301-661): every rank holds 64 / world
    slices of the frames, ONE all-gather (RCCL) of the 64 slice records per iteration.  Strong scaling on a fixed problem (default: the
    embeddings of 64 h = 345 600 frames x 128, 24 speakers); rank 0 also runs the whole problem alone (fa_vbx_refine) and the ELBO
    histories must be equal bit for bit — the sharded run computes the same slice records."""
    from fluidaudio_amd.sharding import VbxShard, all_gather_records, vbx_refine_sharded, vbx_shard_frames
    if 64 % world:
        return {"skipped": f"world size {world} does not divide the 64 slices"}
    T, D, K = int(hours * 5400), 128, 24
    per = -(-T // 64)

    def frames(z):                       # slice z of the problem, the same bytes on whichever rank generates it
        rng = np.random.default_rng(1000 + z)
        n = max(0, min((z + 1) * per, T) - z * per)
        means = np.random.default_rng(999).standard_normal((K, D)) * 30.0 * 0.3
        spk = rng.integers(0, K, n)
        x = means[spk] + 30.0 * 0.1 * rng.standard_normal((n, D))
        init = spk.copy()
        flip = rng.random(n) < 0.1
        init[flip] = rng.integers(0, K, int(flip.sum()))
        return x, init.astype(np.int32)
    phi = np.random.default_rng(998).uniform(0.5, 4.0, D)
    lo, hi = vbx_shard_frames(T, rank, world)
    zn = 64 // world
    parts = [frames(z) for z in range(rank * zn, (rank + 1) * zn)]
    x = np.concatenate([p[0] for p in parts]); init = np.concatenate([p[1] for p in parts])
    assert x.shape[0] == hi - lo
    err = None
    try:                                                  # local set-up: a rank that fails here must not leave the others inside a collective
        shard = VbxShard(x, init, T, K, phi, rank, world, ctx=ctx)
    except Exception as e:  # noqa: BLE001
        err, shard = repr(e), None
    if dist is not None:
        okf = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if okf.item() != 1.0:
            if shard is not None:
                shard.close()
            return {"error": err or "another rank failed to set its shard up"}
    elif err:
        return {"error": err}
    gather = all_gather_records(dist) if dist is not None else (lambda c: c)
    vbx_refine_sharded(shard, gather, 2, 0.0)            # warm-up (RCCL set-up, first launches)
    shard.close()
    shard = VbxShard(x, init, T, K, phi, rank, world, ctx=ctx)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, pi, hard, elbos = vbx_refine_sharded(shard, gather, 20, 1e-4)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    shard.close()
    tt = torch.tensor([wall], dtype=torch.float64, device="cuda")
    same = torch.tensor([1.0], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e = torch.tensor(elbos + [0.0] * (20 - len(elbos)), dtype=torch.float64, device="cuda")
        emin, emax = e.clone(), e.clone()
        dist.all_reduce(emin, op=dist.ReduceOp.MIN); dist.all_reduce(emax, op=dist.ReduceOp.MAX)
        same[0] = float(bool((emin == emax).all()))
    out = {"workload": f"VBx over {T} frames x {D}, {K} speakers, sharded over the frame axis on {world} GPU(s)", "world": world, "iterations": len(elbos),
           "sharded_s": float(tt.item()), "all_ranks_same_elbos": bool(same.item() == 1.0), "scaling": "strong",
           "all_gather_bytes_per_iteration": 64 * (K * (D + 1) + 1) * 8}
    if rank == 0:
        allp = [frames(z) for z in range(64)]
        X = np.concatenate([p[0] for p in allp]); I = np.concatenate([p[1] for p in allp])
        v = fa.VBxClustering(phi, ctx=ctx)
        v.refine(X, I)
        t0 = time.perf_counter()
        one = v.refine(X, I)
        out["single_device_s_incl_host_copies"] = time.perf_counter() - t0
        out["elbos_equal_single_device"] = one.elbos == elbos
        out["hard_labels_equal_single_device"] = bool(np.array_equal(np.asarray(one.hard_clusters[0][lo:hi], np.int32), hard))
    return out


def headline_leg(fa, ctx, torch, dist, rank, world, steps, warmup, hours=8.0):
    """The timed region of the bench: K x (mel over the recording's chunks + fa_offline_cluster on its embeddings), inputs resident."""
    from e2e_inputs import sha256 as sha
    s, gold, aux = e2e_session_for(rank, hours)
    n_chunks15 = int(hours * 3600 / 15)
    d_pcm = synth_pcm(torch, n_chunks15, 99 + rank)
    mel = fa.AudioMelSpectrogram(ctx=ctx)
    plan = mel.plan(np.arange(n_chunks15 + 1, dtype=np.int64) * CHUNK_SAMPLES, layout="mel_major")
    d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
    d_len = torch.empty(n_chunks15, dtype=torch.int32, device="cuda")
    d_emb = torch.from_numpy(s["emb"]).cuda()
    d_rho = torch.from_numpy(s["rho"]).cuda()
    torch.cuda.synchronize()
    n = len(s["emb"])
    # cold start: the first call of a context at this size allocates the linkage workspace (N^2 * 8 B = 15 GB at 43 200 rows)
    t0 = time.perf_counter()
    first = fa.cluster_embeddings(d_emb, d_rho, s["chunks"], s["phi"], ctx=ctx)
    first_call_s = time.perf_counter() - t0
    plan.execute(d_pcm, d_out, d_len, order=False)
    ctx.synchronize()

    def step():
        plan.execute(d_pcm, d_out, d_len, order=False)
        return fa.cluster_embeddings(d_emb, d_rho, s["chunks"], s["phi"], ctx=ctx, intermediates=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    results = [step() for _ in range(steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_elapsed = [elapsed]
    if dist is not None:
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_elapsed = [float(v) for v in every]
        tt = mine.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    # mel share of a step, timed on its own (not part of the timed region above)
    t0 = time.perf_counter()
    plan.execute(d_pcm, d_out, d_len, order=False)
    ctx.synchronize()
    t_mel = time.perf_counter() - t0
    assert int(d_len[0]) == 1501 and bool(torch.isfinite(d_out[n_chunks15 // 2]).all())
    # ---- verification of what was timed
    lab = [np.asarray(r.assignments, np.int32) for r in results]
    same_every_step = all(np.array_equal(lab[0], x) for x in lab[1:])
    pure = len(set(zip(s["spk"].tolist(), lab[-1].tolist()))) == s["speakers"]
    digest = None
    if gold is not None:
        chk = fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=ctx, intermediates=True)   # host pointers + copies of the intermediates
        digest = {"file": "tests/golden/e2e_8h.json",
                  "assignments": sha(lab[-1]) == gold["assignments_sha256"],
                  "centroids_1e-9": bool(results[-1].centroids.shape == aux["centroids"].shape and np.allclose(results[-1].centroids, aux["centroids"], rtol=0, atol=1e-9)),
                  "ahc_labels": sha(np.asarray(chk.initial_clusters, np.int32)) == gold["ahc_labels_sha256"],
                  "vbx_hard_labels": sha(np.asarray(chk.info["vbx_hard"], np.int32)) == gold["vbx_hard_sha256"],
                  "vbx_iterations": int(chk.info["vbx_iterations"]) == gold["vbx_iterations"],
                  "elbos_1e-9": bool(np.allclose(chk.info["elbos"], gold["vbx_elbos"], rtol=1e-9, atol=0)),
                  "host_pointer_call_equals_device_pointer_call": bool(np.array_equal(np.asarray(chk.assignments, np.int32), lab[-1])),
                  "cpu_seconds_1_core_when_generated": gold["cpu_seconds_1_core"]}
        digest["all"] = all(v for k, v in digest.items() if isinstance(v, bool))
    ok = torch.tensor([1.0 if (pure and same_every_step and (digest is None or digest["all"])) else 0.0], device="cuda")
    if dist is not None:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    # ---- the dominant kernel: one ahc_round_t launch per round; period = merge-phase HIP-event time / rounds
    rounds = sum(r.info["ahc"]["rounds"] for r in results)
    merge_ms = sum(r.info["ahc"]["merge_ms"] for r in results)
    init_ms = sum(r.info["ahc"]["init_ms"] for r in results) / len(results)
    period_us = 1e3 * merge_ms / max(1, rounds)
    npad = (n + 255) // 256 * 256
    round_bytes = 3 * npad * 8                      # two operand rows read + one row written, fp64, per merge
    ach = round_bytes / (period_us * 1e-6) / 1e9
    traffic, tsrc = measured_traffic_of("*_ahc_round_pmc.json", AHC_SOURCES)
    gram_flop = 2.0 * npad * npad * s["emb"].shape[1] / 2.0      # tiles on and below the diagonal only
    stage = {k: float(np.mean([r.timings[k] for r in results])) for k in results[0].timings}
    out = {"elapsed": elapsed, "per_rank_audio_hours_per_s": [hours * steps / e for e in per_rank_elapsed], "hours": hours, "first_call_s": first_call_s, "first_call_note": "fresh context: includes the hipMalloc of the 15 GB linkage workspace",
           "mel_s": t_mel, "mel_chunks": n_chunks15, "embeddings": n, "stages_s": stage, "clusters_found": int(results[-1].centroids.shape[0]),
           "speakers_true": s["speakers"], "labels_match_speakers": bool(pure), "identical_every_step": bool(same_every_step),
           "e2e_equals_reference_digest": None if digest is None else bool(digest["all"]), "digest_checks": digest, "all_ranks_ok": bool(ok.item() > 0.5),
           "ahc": {"rounds_per_step": rounds / len(results), "us_per_round": period_us, "init_ms": init_ms, "merge_ms": merge_ms / len(results),
                   "windows": results[-1].info["ahc"]["windows"], "exact_fallback": results[-1].info["ahc"]["exact_fallback"],
                   "gram": {"bound": "mfma", "achieved": gram_flop / 1e12 / max(init_ms * 1e-3, 1e-9), "peak": 78.6, "unit": "TFLOP/s",
                            "note": "fp64 matrix-core start-up (ahc_gram_mfma): flops of the half Gram matrix / the WHOLE start-up time (transpose, norms, Gram, row minima, records)"}},
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                        "kernel": "ahc_round_t<false>", "launch_period_us": period_us, "launches_per_step": rounds / len(results),
                        "algorithmic_bytes_per_launch": round_bytes,
                        "note": "serial-chain-bound, not bandwidth-bound: N - 1 dependent merges, each = kernel boundary + two dependent memory round trips + "
                                "two reductions (DESIGN.md §3.3); the period is the HIP-event time of the merge phase / launches, i.e. launch duration + boundary"}}
    del d_out, d_pcm, d_emb, d_rho
    return out


def e2e_hard_leg(fa, ctx, torch, steps=3):
    """The hard 8 h session (sigma = 0.041, tests/golden/e2e_8h_s0p041.json): AHC at threshold 0.6 leaves hundreds of clusters, VBx runs over
    S = that many speakers and prunes them to 12, the constrained assignment moves thousands of embeddings (VBxClustering.swift:301-661,
    OfflineDiarizerManager.swift:613-691) — the session where vbx_estep / vbx_gt_rho / the Hungarian stage see a large S; the headline session
    (sigma 0.03) hands VBx 12 clusters.  Timed like the headline (inputs resident, labels returned), every step digest-checked."""
    from e2e_inputs import e2e_session, input_digest, sha256
    gp = os.path.join(ROOT, "tests", "golden", "e2e_8h_s0p041.json")
    with open(gp) as f:
        gold = json.load(f)
    s = e2e_session(gold["hours"], gold["speakers"], sigma=gold.get("sigma", 0.041))
    if input_digest(s) != gold["input_sha256"]:
        return {"skipped": "this numpy regenerates different input bytes: the digests do not apply"}
    d_emb = torch.from_numpy(s["emb"]).cuda()
    d_rho = torch.from_numpy(s["rho"]).cuda()
    chk = fa.cluster_embeddings(d_emb, d_rho, s["chunks"], s["phi"], ctx=ctx, intermediates=True)       # warm-up + the intermediates for the digests
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = [fa.cluster_embeddings(d_emb, d_rho, s["chunks"], s["phi"], ctx=ctx, intermediates=False) for _ in range(steps)]
    wall = (time.perf_counter() - t0) / steps
    ok = all(sha256(np.asarray(r.assignments, np.int32)) == gold["assignments_sha256"] for r in res)
    digest = {"assignments_every_step": bool(ok), "ahc_labels": sha256(np.asarray(chk.initial_clusters, np.int32)) == gold["ahc_labels_sha256"],
              "vbx_hard_labels": sha256(np.asarray(chk.info["vbx_hard"], np.int32)) == gold["vbx_hard_sha256"],
              "vbx_iterations": int(chk.info["vbx_iterations"]) == gold["vbx_iterations"],
              "elbos_1e-9": bool(np.allclose(chk.info["elbos"], gold["vbx_elbos"], rtol=1e-9, atol=0))}
    stage = {k: float(np.mean([r.timings[k] for r in res])) for k in res[0].timings}
    return {"hours": gold["hours"], "sigma": gold.get("sigma"), "embeddings": len(s["emb"]), "ahc_clusters": int(chk.info["initial_clusters"]),
            "vbx_iterations": int(chk.info["vbx_iterations"]), "clusters_found": int(res[-1].centroids.shape[0]), "seconds_per_recording": wall,
            "audio_hours_per_s": gold["hours"] / wall, "stages_s": stage, "equals_reference_digest": all(digest.values()), "digest_checks": digest,
            "cpu_seconds_1_core_when_generated": gold.get("cpu_seconds_1_core")}


def cpu_e2e_baseline(hours=1.0):
    """The same path on this box's host cores, 1 thread, on a bounded sample (`hours` of audio): oracle mel on its chunks + the
    REFERENCE's own linkage build (oracle/_ref) + the C restatements of VBx / centroids / Hungarian.  The clustering cost grows like
    N^2, so the rate of a shorter recording OVERSTATES what the CPU reaches on 8 h (committed: 607 s for the 8 h linkage alone)."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from e2e_inputs import e2e_session
    s = e2e_session(hours, 12, seed=5)
    rng = np.random.default_rng(1234)
    t = np.arange(CHUNK_SAMPLES) / 16000.0
    chunk = (rng.uniform(-1, 1, CHUNK_SAMPLES) * 0.1 + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    oracle.mel_flat(chunk[:16000])
    n_chunks = int(hours * 240)
    timed = min(n_chunks, 64)
    t0 = time.perf_counter()
    for _ in range(timed):
        oracle.mel_flat(chunk)
    t_mel = (time.perf_counter() - t0) * n_chunks / timed
    t0 = time.perf_counter()
    oracle.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"])
    t_cl = time.perf_counter() - t0
    return {"value": hours / (t_mel + t_cl), "unit": "audio_hours/s", "cores": 1, "kind": "port",
            "sample": f"{hours:g} h recording ({len(s['emb'])} embeds), 1 of {os.cpu_count()} cores: mel port {t_mel:.1f} s + linkage (reference build) and VBx/assignment ports {t_cl:.1f} s",
            "kind_note": "only the linkage is the reference's own code (its FastClusterWrapper C++ compiled by oracle/Makefile); mel, VBx, centroids and Hungarian are "
                         "CPU restatements (ports) of the Swift sources — the Swift/Accelerate path itself cannot run here",
            "sample_long": f"{hours:g} h recording ({len(s['emb'])} embeddings): clustering {t_cl:.1f} s = the reference's FastClusterWrapper build (oracle/_ref) + C "
                      f"restatements of VBx / centroids / Hungarian; mel {t_mel:.1f} s = oracle computeFlat restatement on {n_chunks} chunks ({timed} timed, scaled); "
                      f"1 of {os.cpu_count()} host cores; Swift/Accelerate itself cannot run on this box",
            "mel_s": t_mel, "cluster_s": t_cl,
            "full_size_note": "measured at the full size in this run (--cpu-full)" if hours >= 8.0 else
                              "NOT measured in this run (a stored figure): 8 h (43 200 embeddings) on one core of the build container took linkage 607 s + VBx/assignment 2 s "
                              "(tests/golden/e2e_8h.json) + mel ~430 s at this rate => ~0.008 audio-hours/s; `bench.py --cpu-full` times it here"}


def mel_leg(fa, ctx, torch, dist, rank, world, B, steps, warmup, clock_warm_s, measure_aligned=True):
    """BASELINE configs[1]: one fa_mel_execute_dev over B x 15 s chunks resident in HBM per step; HIP events per launch on the context's stream."""
    stream = torch.cuda.ExternalStream(ctx.stream)
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
    t_warm = time.perf_counter() + max(0.0, clock_warm_s)   # set-up: bring the device to sustained clocks (not timed, not a step)
    while time.perf_counter() < t_warm:
        plan.execute(d_pcm, d_out, d_len, order=False)
        ctx.synchronize()
    for _ in range(warmup):
        plan.execute(d_pcm, d_out, d_len, order=False)
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(steps):
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
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    kernel_ms_avg = float(np.mean(kernel_ms))
    assert int(d_len[0]) == 1501 and bool(torch.isfinite(d_out[B // 2]).all())
    hours = world * B * 15.0 / 3600.0
    value = hours * steps / elapsed
    ach = B * MEL_BYTES_PER_CHUNK / (kernel_ms_avg * 1e-3) / 1e9
    traffic, traffic_source = measured_traffic()
    # what the unaligned output rows cost: the same launch with frame_stride 1504 (rows of 6 016 bytes = 47 x 128-byte lines) instead
    # of the reference's 1501 (6 004 bytes: every row starts inside a line its neighbour also writes)
    aligned = None
    try:
        if not measure_aligned:
            raise RuntimeError("skipped (profiling run)")
        plan_a = mel.plan(offsets, layout="mel_major", frame_stride=1504)
        d_out_a = torch.empty(plan_a.out_shape(), dtype=torch.float32, device="cuda")
        for _ in range(10):
            plan_a.execute(d_pcm, d_out_a, d_len, order=False)
        ctx.synchronize()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        na = max(20, steps // 4)
        ea.record(stream)
        for _ in range(na):
            plan_a.execute(d_pcm, d_out_a, d_len, order=False)
        eb.record(stream)
        ctx.synchronize()
        ms_a = ea.elapsed_time(eb) / na
        same = bool(torch.equal(d_out_a[:, :, :1501], d_out))
        aligned = {"frame_stride": 1504, "kernel_ms_avg": ms_a, "speedup_vs_1501": kernel_ms_avg / ms_a, "same_values": same,
                   "note": "row stride of 47 whole 128-byte lines: measures the cost of the 1.2x write amplification of the reference's [128, 1501] layout"}
        del d_out_a
        plan_a.close()
    except Exception as e:  # noqa: BLE001
        aligned = {"error": repr(e)}
    return {"workload": "BASELINE configs[1]: batched STFT->mel, 1024 x 15 s 16 kHz chunks per GPU, NeMo config (n_fft 512, hop 160, win 400, 128 mels, "
                        "preemph 0.97), output [B,128,1501] fp32, inputs resident in HBM", "chunks_per_gpu": B, "steps": steps, "warmup": warmup,
            "audio_hours_per_s": value, "realtime_factor": value * 3600.0, "ms_per_step": 1e3 * elapsed / steps, "scaling": "weak", "aligned_rows": aligned,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "mel_kernel_v4<MEL_MAJOR>", "kernel_ms_avg": kernel_ms_avg, "kernel_ms_min": float(np.min(kernel_ms)),
                         "algorithmic_bytes_per_launch": B * MEL_BYTES_PER_CHUNK,
                         "note": "VALU / LDS-issue bound rather than HBM bound (DESIGN.md §3.1)"}}



# ---------------------------------------------------------------------------------------------------------------------------------------
# What is printed.  The driver parses the LAST stdout line and keeps the last 8 KB of stdout; a line that grew to 25 KB (round 5) was not
# parsed at all.  So: every leg is printed as its own line the moment it is done ({"leg": name, "result": {...}}), all of them are
# written to bench_legs.json, then one {"summary": ...} line of the scalars worth a glance, then the RESULT line — the contract's keys,
# scalars only, strings <= 120 characters, < 4 KB (tests/test_bench_line.py holds it to that on a stored full record).
RESULT_LINE_MAX_BYTES = 4096
_STR_MAX = 120


def _pick(d, *path):
    for k in path:
        d = d.get(k) if isinstance(d, dict) else None
    return d


def _num(v, digits=6):
    """Scalars of the printed lines: 6 significant digits are what a reader compares; NaN / inf have no strict-JSON form."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v[:_STR_MAX] if isinstance(v, str) else v
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if math.isfinite(v) else None
    if isinstance(v, (list, tuple)):
        return [_num(x, digits) for x in v]
    try:
        return _num(float(v), digits)
    except (TypeError, ValueError):
        return str(v)[:_STR_MAX]


class LegRecord(dict):
    """The full record of a run.  Assigning a leg (a dict) prints it at once; the contract's own keys are set through update()."""

    def __init__(self, rank):
        super().__init__()
        self.rank = rank

    def __setitem__(self, name, result):
        super().__setitem__(name, result)
        if isinstance(result, dict):
            emit_leg(name, result, self.rank)


def _strict_tree(v):
    """A leg as strict JSON: non-finite floats (a 0 / 0 of a leg that measured nothing) become null instead of the bare NaN json.dumps would print."""
    if isinstance(v, dict):
        return {str(k): _strict_tree(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_strict_tree(x) for x in v]
    if isinstance(v, (np.floating, float)):
        return float(v) if math.isfinite(v) else None
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.bool_):
        return bool(v)
    return v


def emit_leg(name, result, rank=0):
    """One line per leg, as soon as it exists (a later crash keeps the earlier legs; the driver's 8 KB tail shows the last ones)."""
    if rank == 0:
        print(json.dumps({"leg": name, "result": _strict_tree(result)}, allow_nan=False, default=str), flush=True)


def summary_of(full):
    """The scalars of the legs a reader wants beside the headline (each from the leg of the same name in bench_legs.json)."""
    batch = full.get("e2e_8h_batch") if isinstance(full.get("e2e_8h_batch"), dict) else {}
    s = {
        "e2e_equals_reference_digest": full.get("e2e_equals_reference_digest"),
        "e2e_digest_means": "reference-build linkage + restated VBx/centroids/Hungarian (their parity is unpinned, DESIGN.md section 2)",
        "e2e_per_rank_audio_hours_per_s": _pick(full, "e2e_8h", "per_rank_audio_hours_per_s"), "e2e_all_ranks_ok": _pick(full, "e2e_8h", "all_ranks_ok"),
        "ahc_50k_seconds": _pick(full, "ahc_50k", "seconds"), "ahc_50k_bit_exact_vs_reference_digest": _pick(full, "ahc_50k", "bit_exact_vs_reference_digest"),
        "ahc_50k_us_per_round": _pick(full, "ahc_50k", "us_per_round"), "ahc_50k_init_ms": _pick(full, "ahc_50k", "device_init_ms"),
        "e2e_us_per_round": _pick(full, "e2e_8h", "ahc", "us_per_round"), "e2e_ahc_init_ms": _pick(full, "e2e_8h", "ahc", "init_ms"),
        "startup_tflops_fp64": _pick(full, "e2e_8h", "ahc", "gram", "achieved"),
        "ahc_tied_over_tie_free": _pick(full, "ahc_ties", "tied_over_tie_free"), "ahc_tied_seconds": _pick(full, "ahc_ties", "tied_seconds"),
        "ahc_tied_equals_reference_digest": _pick(full, "ahc_ties", "equals_reference_digest"),
        "mel_frac": _pick(full, "mel", "roofline", "frac"), "mel_realtime_factor": _pick(full, "mel", "realtime_factor"),
        "mel_audio_hours_per_s": _pick(full, "mel", "audio_hours_per_s"),
        "ctc_frac": _pick(full, "ctc", "roofline", "frac"), "ctc_ids_exact": _pick(full, "ctc", "ids_exact"), "ctc_scaling": _pick(full, "ctc", "scaling"),
        "ctc_matrices_per_s": _pick(full, "ctc", "matrices_per_s"), "ctc_audio_hours_per_s": _pick(full, "ctc", "audio_hours_per_s"),
        "ctc_matrices_per_rank": _pick(full, "ctc", "matrices_per_rank"),
        "ctc_fp16_frac": _pick(full, "ctc_fp16", "roofline", "frac"), "ctc_v1025_frac": _pick(full, "ctc_v1025", "roofline", "frac"),
        "ctc_v1025_fp16_frac": _pick(full, "ctc_v1025_fp16", "roofline", "frac"),
        "tdt_frac": _pick(full, "tdt", "roofline", "frac"), "tdt_4096_frac": _pick(full, "tdt_4096", "roofline", "frac"),
        "tdt_4096_fp16_frac": _pick(full, "tdt_4096_fp16", "roofline", "frac"), "tdt_ids_equal_cpu": _pick(full, "tdt", "ids_equal_cpu_restatement_all_chunks"),
        "resample_frac": {k.split("->")[0]: _pick(v, "roofline", "frac") for k, v in (full.get("resample") or {}).items() if isinstance(v, dict) and "->" in k},
        "beam_us_per_frame_step": _pick(full, "beam_search", "us_per_frame_step"), "beam_sclk_mhz": _pick(full, "beam_search", "sclk_mhz_before_after"),
        "e2e_batch_audio_hours_per_s": {k: _pick(v, "audio_hours_per_s") for k, v in batch.items() if k.startswith("x") and isinstance(v, dict)},
        "e2e_batch_us_per_round": {k: _pick(v, "us_per_round") for k, v in batch.items() if k.startswith("x") and isinstance(v, dict)},
        "e2e_batch_equal_single_calls": {k: _pick(v, "equal_single_calls") for k, v in batch.items() if k.startswith("x") and isinstance(v, dict)},
        "e2e_16x1h_audio_hours_per_s": _pick(full, "e2e_16x1h", "audio_hours_per_s"),
        "e2e_hard_audio_hours_per_s": _pick(full, "e2e_8h_hard", "audio_hours_per_s"),
        "vbx_sharded_all_ranks_same_elbos": _pick(full, "vbx_sharded", "all_ranks_same_elbos"),
        "errors": sorted(k for k, v in full.items() if isinstance(v, dict) and "error" in v),
    }
    return {k: (_num(v) if not isinstance(v, dict) else {kk: _num(vv) for kk, vv in v.items()}) for k, v in s.items()}


def result_line_of(full):
    """The one line the driver parses: the contract's keys + `roofline` + `cpu_baseline`, scalars only."""
    batch = full.get("e2e_8h_batch") if isinstance(full.get("e2e_8h_batch"), dict) else {}
    cfg = full.get("config") or {}
    roof = full.get("roofline") or {}
    cpu = full.get("cpu_baseline")
    out = {k: _num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                           "dtype", "data")}
    out["config"] = {
        "workload": _num(cfg.get("workload")),
        "hours_per_step_per_gpu": _num(cfg.get("hours_per_step_per_gpu")), "embeddings_per_recording": _num(cfg.get("embeddings_per_recording")),
        "realtime_factor": _num(cfg.get("realtime_factor")), "parallelism": _num(cfg.get("parallelism")),
        "e2e_equals_reference_digest": full.get("e2e_equals_reference_digest"),
        "ahc_50k_seconds": _num(_pick(full, "ahc_50k", "seconds")), "ahc_50k_bit_exact_vs_reference_digest": _pick(full, "ahc_50k", "bit_exact_vs_reference_digest"),
        "mel_realtime_factor": _num(_pick(full, "mel", "realtime_factor")), "mel_roofline_frac": _num(_pick(full, "mel", "roofline", "frac")),
        "ctc_roofline_frac": _num(_pick(full, "ctc", "roofline", "frac")), "ctc_ids_exact": _pick(full, "ctc", "ids_exact"),
        "batch_x8_audio_hours_per_s": _num(_pick(batch, "x8", "audio_hours_per_s")),
    }
    if (full.get("n_gpus") or 1) > 1:
        # N > 1: the single-GPU legs did not run; their places carry what a scaling record needs from the legs the driver drops — the strong-scaled
        # configs[3] rate, the slowest rank's own end-to-end rate, and that the one leg with a collective agreed across ranks
        for k in ("ahc_50k_seconds", "ahc_50k_bit_exact_vs_reference_digest", "batch_x8_audio_hours_per_s", "mel_realtime_factor"):
            out["config"].pop(k)
        per_rank = _pick(full, "e2e_8h", "per_rank_audio_hours_per_s") or [None]
        out["config"].update({
            "e2e_all_ranks_ok": _pick(full, "e2e_8h", "all_ranks_ok"),
            "e2e_slowest_rank_audio_hours_per_s": _num(min((v for v in per_rank if v is not None), default=None)),
            "ctc_matrices_per_s_strong_scaled": _num(_pick(full, "ctc", "matrices_per_s")),
            "vbx_sharded_all_ranks_same_elbos": _pick(full, "vbx_sharded", "all_ranks_same_elbos"),
        })
    out["roofline"] = {k: _num(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_period_us", "launches_per_step",
                                                        "algorithmic_bytes_per_launch")}
    if isinstance(cpu, dict):
        out["cpu_baseline"] = {k: _num(cpu.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "mel_s", "cluster_s") if k in cpu} if "error" not in cpu \
            else {"error": _num(cpu["error"])}
    else:
        out["cpu_baseline"] = None            # N > 1 (rank 0 of a multi-rank run does not time the CPU side) or --skip-cpu
    out["legs_file"] = "bench_legs.json"
    return out


def finish(full, rank):
    """Rank 0: bench_legs.json (everything), the summary line, the result line (last)."""
    if rank != 0:
        return
    for dest in (os.path.join(ROOT, "bench_legs.json"), os.path.join(ROOT, "gpurun_out", "bench_legs.json")):
        if os.path.isdir(os.path.dirname(dest)):
            try:
                with open(dest, "w") as f:
                    json.dump(_strict_tree(full), f, allow_nan=False, default=str)
            except OSError as e:
                print(f"bench.py: could not write {dest}: {e}", file=sys.stderr)
    print(json.dumps({"summary": summary_of(full)}, allow_nan=False), flush=True)
    text = json.dumps(result_line_of(full), allow_nan=False)
    assert len(text) < RESULT_LINE_MAX_BYTES, len(text)
    print(text, flush=True)

def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this script, one per GPU, under torch.distributed.run
    (the form the driver uses for N > 1) and hand its exit code back.  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def ranks_seen(torch, dist, device):
    """How many ranks the process group really joined: the sum of a one from every rank."""
    one = torch.ones(1, dtype=torch.int64, device=device) if device is not None else torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    return int(one.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed steps; one step = one 8 h recording featurized + clustered (~0.3 s)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hours", type=float, default=8.0, help="length of the recording of a step (BASELINE configs[4]: 8)")
    ap.add_argument("--mel-steps", type=int, default=200, help="launches of the configs[1] mel leg")
    ap.add_argument("--mel-warmup", type=int, default=20)
    ap.add_argument("--clock-warm-s", type=float, default=0.3,
                    help="mel leg: seconds of the same launches before its warm-up steps: the device leaves idle clocks only after ~30 ms "
                         "of sustained load (launch time falls from 0.95 to 0.68 ms over the first ~40 launches, scripts/mel_variance.py)")
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_GPU, help="mel leg: 15 s chunks per GPU per launch (BASELINE config: 1024)")
    ap.add_argument("--skip-mel", action="store_true")
    ap.add_argument("--skip-ahc", action="store_true")
    ap.add_argument("--skip-ctc", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU side on the full --hours session instead of the bounded 1 h sample (8 h: ~20 minutes on one core)")
    ap.add_argument("--skip-e2e", action="store_true", help="skip the 16 x 1 h leg (the headline itself cannot be skipped)")
    ap.add_argument("--skip-beam", action="store_true")
    ap.add_argument("--skip-resample", action="store_true")
    ap.add_argument("--in-flight", action="store_true", help="also run round 3's four-chains-in-flight leg (queue-dependent; superseded by e2e_8h_batch)")
    ap.add_argument("--vbx-sharded", action="store_true", help="run the sharded-VBx leg at N = 1 too (it always runs at N > 1)")
    ap.add_argument("--only-mel", action="store_true", help="profiling helper: the configs[1] mel leg alone, printed as a reduced line")
    ap.add_argument("--ctc-matrices", type=int, default=10000)
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, form the process group, all-reduce a one per rank, print what was seen; no device work")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))                          # `python bench.py --gpus N` starts its own N ranks
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # one rank per GPU is the contract: a launcher that started a different number of ranks than --gpus says is a mistake, not a warning
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("FA_BENCH_BACKEND", "nccl")          # "gloo": a rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.launch_check:                                      # control flow only (runs without a GPU): rendezvous + one all-reduce
            dist.init_process_group("gloo" if not torch.cuda.is_available() else backend, rank=rank, world_size=world)
            seen = ranks_seen(torch, dist, torch.device("cuda", local_rank % torch.cuda.device_count()) if torch.cuda.is_available() and backend == "nccl" else None)
            if rank == 0:
                print(json.dumps({"launch_check": True, "n_gpus": seen, "world_size_env": world, "backend": dist.get_backend()}))
            dist.barrier()
            dist.destroy_process_group()
            return
        have = torch.cuda.device_count()
        if backend == "nccl" and have < world:
            if rank == 0:
                print(f"bench.py: --gpus {world} needs {world} visible GPUs, this box has {have} (FA_BENCH_BACKEND=gloo rehearses the control flow "
                      "with several ranks per GPU)", file=sys.stderr)
            sys.exit(2)
        if backend != "nccl":
            local_rank %= max(have, 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        seen = ranks_seen(torch, dist, torch.device("cuda", local_rank) if backend == "nccl" else None)
        if seen != world:
            raise RuntimeError(f"the process group saw {seen} ranks, WORLD_SIZE says {world}")
    else:
        if args.launch_check:
            print(json.dumps({"launch_check": True, "n_gpus": 1, "world_size_env": 1, "backend": None}))
            return
        torch.cuda.set_device(local_rank)
    import fluidaudio_amd as fa
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

    ctx = fa.default_context(local_rank)
    solo = world == 1
    if args.only_mel:
        m = mel_leg(fa, ctx, torch, dist, rank, world, args.chunks, args.mel_steps, args.mel_warmup, args.clock_warm_s, measure_aligned=False)
        if rank == 0:
            print(json.dumps({"metric": "mel leg only (profiling helper)", "mel": m}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------- the timed region: K steps of configs[4]
    h = headline_leg(fa, ctx, torch, dist, rank, world, args.steps, args.warmup, args.hours)
    elapsed = h.pop("elapsed")
    value = world * args.hours * args.steps / elapsed
    roof = h.pop("roofline")
    line = LegRecord(rank)
    line.update({
        "metric": "audio hours/sec featurized+clustered per node; AHC wall-clock @ 50k x 256 embeds (ahc_50k.seconds)",
        "value": value, "unit": "audio_hours/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        # strings of the result line stay <= 120 characters (the driver cuts at 128); the long form of the workload is DESIGN.md section 5
        "config": {"workload": f"configs[4]: per GPU one {args.hours:g} h 16 kHz recording -> mel -> {int(args.hours * 5400)} embeds -> AHC+VBx+assignment, HBM-resident",
                   "hours_per_step_per_gpu": args.hours, "embeddings_per_recording": h["embeddings"], "realtime_factor": value * 3600.0,
                   "parallelism": f"dp{world}"},
        "e2e_equals_reference_digest": h["e2e_equals_reference_digest"],
        "roofline": roof,
    })
    line["e2e_8h"] = h
    torch.cuda.empty_cache()
    if solo and not args.skip_cpu and rank == 0:
        try:
            cpu = cpu_e2e_baseline(args.hours if args.cpu_full else 1.0)
            cpu["stages"] = cpu_baselines()
            line["cpu_baseline"] = cpu
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"error": repr(e)}
    if not args.skip_mel:
        try:
            line["mel"] = mel_leg(fa, ctx, torch, dist, rank, world, args.chunks, args.mel_steps, args.mel_warmup, args.clock_warm_s)
        except Exception as e:  # noqa: BLE001
            line["mel"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if not args.skip_ctc:
        try:
            r = ctc_leg(fa, ctx, torch, dist, rank, world, args.ctc_matrices)
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)}
        line["ctc"] = r
        torch.cuda.empty_cache()
    if not solo or args.vbx_sharded:   # the one leg with a data-path collective (§8e row 4); at N = 1 only on request
        try:
            r = vbx_sharded_leg(fa, ctx, torch, dist, rank, world)
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)}
        line["vbx_sharded"] = r
        torch.cuda.empty_cache()
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if solo and not args.skip_mel:
        try:
            line["mel_single_10s"] = mel_single_leg(fa, ctx)
        except Exception as e:  # noqa: BLE001
            line["mel_single_10s"] = {"error": repr(e)}
    if solo and not args.skip_ctc:
        try:
            line["ctc_fp16"] = ctc_leg(fa, ctx, torch, None, 0, 1, args.ctc_matrices, dtype="f16")
            for name, dt in (("ctc_v1025", "f32"), ("ctc_v1025_fp16", "f16")):   # the shape the model really emits: rows of any alignment (head + body + tail path)
                line[name] = ctc_leg(fa, ctx, torch, None, 0, 1, min(args.ctc_matrices, 4000), dtype=dt, vocab=1025)
        except Exception as e:  # noqa: BLE001
            line["ctc_fp16"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            line["tdt"] = tdt_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["tdt"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        for name, kw in (("tdt_4096_fp16", dict(B=4096, dtype="float16")), ("tdt_4096", dict(B=4096))):   # the batch is the parallel axis: four chunks per SIMD
            try:                                                                                           # instead of one (101 GB of fp16 / 203 GB of fp32 logits resident)
                line[name] = tdt_leg(fa, ctx, torch, **kw)
            except Exception as e:  # noqa: BLE001
                line[name] = {"error": repr(e)}
            torch.cuda.empty_cache()
    if solo and not args.skip_resample:
        try:
            line["resample"] = resample_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["resample"] = {"error": repr(e)}
        torch.cuda.empty_cache()
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
        try:
            line["ahc_ties"] = ahc_ties_leg(fa, ctx)
        except Exception as e:  # noqa: BLE001
            line["ahc_ties"] = {"error": repr(e)}
    if solo and not args.skip_e2e:
        torch.cuda.empty_cache()
        try:
            line["e2e_16x1h"] = e2e_many_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["e2e_16x1h"] = {"error": repr(e)}
        try:
            line["e2e_8h_batch"] = e2e_batch_leg(fa, ctx)
        except Exception as e:  # noqa: BLE001
            line["e2e_8h_batch"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            line["e2e_8h_hard"] = e2e_hard_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["e2e_8h_hard"] = {"error": repr(e)}
        ctx.trim()
        torch.cuda.empty_cache()
        if args.in_flight:
            # round 3's throughput form (four host threads, a context and a chain each).  Its rate depends on which hardware queues the process's
            # streams landed on (89 audio-hours/s in a fresh leg order, 32 after the batch leg above has created a helper stream: profiles/
            # r04_bench_v5.json / _v6.json), which is why the uniform batches (e2e_8h_batch) replaced it as the serving path; kept on request.
            try:
                line["e2e_8h_x4_in_flight"] = e2e_in_flight_leg(fa, torch)
            except Exception as e:  # noqa: BLE001
                line["e2e_8h_x4_in_flight"] = {"error": repr(e)}
            torch.cuda.empty_cache()
    if solo and not args.skip_beam:
        torch.cuda.empty_cache()
        try:
            line["beam_search"] = beam_leg(fa, ctx, torch)
        except Exception as e:  # noqa: BLE001
            line["beam_search"] = {"error": repr(e)}
    finish(line, rank)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
