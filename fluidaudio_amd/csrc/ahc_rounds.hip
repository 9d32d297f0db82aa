// ahc_rounds.hip — the filter-based linkage of ONE problem: entry kernels of the round, set-up / replay / finish, fa::ahc_run_device (ahc_ws.h: the map).
#include "ahc_round_body.h"

using namespace fa_ahc;

namespace {
// Exact heights from the stored centroids, the reference's summation order, then sqrt
// (cluster_result::sqrt, FastClusterWrapper.cpp:128-130).
__global__ void ahc_heights(Ws w) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= w.N - 1) return;
    double *z = w.Z + static_cast<size_t>(s) * 4;
    const double *ca = w.C + static_cast<size_t>(z[0]) * w.d, *cb = w.C + static_cast<size_t>(z[1]) * w.d;
    double sum = 0.0;
    for (int k = 0; k < w.d; ++k) {
        const double diff = __dsub_rn(ca[k], cb[k]);
        sum = __dadd_rn(sum, __dmul_rn(diff, diff));
    }
    if (sum != sum) w.flags[0] = 1;
    z[2] = __dsqrt_rn(sum);
}

}  // namespace

namespace fa_ahc {
void window_counter_init(WinCounters (&c)[4]) { for (auto &x : c) { x.stale_key = ~0ULL; x.ncand = 0; x.npairs = 0; } }

fa_status prob_check_shape(fa_ctx *ctx, size_t N, size_t d) {
    const size_t Np = (N + kBlk - 1) / kBlk * kBlk;
    if (Np / kBlk > static_cast<size_t>(kMaxBlocks)) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: N too large for the resident distance matrix");
    if (d * sizeof(double) > 60 * 1024) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: dimension too large for the LDS centroid buffer");
    return FA_SUCCESS;
}

// binds the workspace at `base`, uploads the initial state and runs the start-up kernels (matrix, row minima, records, eps)
void prob_bind(Prob &p, char *base) {
    const size_t N = p.N, d = p.d, Np = p.Np;
    const Layout &L = p.L;
    p.base = base;
    Ws &w = p.w;
    w = Ws{};
    w.state = reinterpret_cast<AhcState *>(base + L.state);
    w.cnt = reinterpret_cast<WinCounters *>(base + L.cnt);
    w.flags = reinterpret_cast<int32_t *>(base + L.flags);
    w.prof = reinterpret_cast<unsigned long long *>(base + L.prof);
    w.recA = reinterpret_cast<RecA *>(base + L.reca);
    w.recI = reinterpret_cast<int4 *>(base + L.reci);
    w.recS = reinterpret_cast<RecS *>(base + L.recs);
    w.recP = reinterpret_cast<RecP *>(base + L.recp);
    w.row = reinterpret_cast<RowSt *>(base + L.row);
    w.e2 = reinterpret_cast<double *>(base + L.e2);
    w.node = reinterpret_cast<int32_t *>(base + L.node);
    w.sizes = reinterpret_cast<double *>(base + L.sizes);
    w.Z = reinterpret_cast<double *>(base + L.z);
    w.cand = reinterpret_cast<int2 *>(base + L.cand);
    w.pairs = reinterpret_cast<int4 *>(base + L.pairs);
    w.C = reinterpret_cast<double *>(base + L.c);
    w.XT = reinterpret_cast<double *>(base + L.xt);
    w.M = reinterpret_cast<double *>(base + L.m);
    w.N = static_cast<int32_t>(N); w.Np = static_cast<int32_t>(Np); w.d = static_cast<int32_t>(d); w.nblk = static_cast<int32_t>(Np / (static_cast<size_t>(kBlk) * p.cpt));
}

fa_status prob_setup(fa_ctx *ctx, Prob &p, char *base) {
    prob_bind(p, base);
    const size_t N = p.N, d = p.d, Np = p.Np;
    const Layout &L = p.L;
    Ws &w = p.w;
    const int dev_mode = p.mode == FA_AHC_MODE_EXACT ? FA_AHC_MODE_EXACT : FA_AHC_MODE_AUTO;
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.C, p.d_data, sizeof(double) * N * d, hipMemcpyDeviceToDevice, ctx->stream));
    startup_filter(ctx->stream, w, L, base, dev_mode, p.d_data, N, Np, d);   // ahc_startup.hip: state, rows, transpose, matrix, row minima, eps
    if (p.cpt == 4) hipLaunchKernelGGL(ahc_records<4>, dim3(w.nblk, 2), dim3(kBlk), 0, ctx->stream, w);  // window counts need eps
    else if (p.cpt == 2) hipLaunchKernelGGL(ahc_records<2>, dim3(w.nblk, 2), dim3(kBlk), 0, ctx->stream, w);
    else hipLaunchKernelGGL(ahc_records<1>, dim3(w.nblk, 2), dim3(kBlk), 0, ctx->stream, w);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

// p.h holds the state after a replay of the round graph: finished, failed, or to be switched to exact rows
fa_status prob_after_replay(fa_ctx *ctx, Prob &p) {
    p.h.rounds = p.h.rounds32;   // the round counter travels in the hot state
    const AhcState &h = p.h;
    if (h.error == 1) { p.active = false; return p.st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance"); }
    if (h.error) { p.active = false; return p.st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: internal selection failure (%d)", h.error); }
    if (h.done) { p.active = false; return FA_SUCCESS; }
    if (h.halt && h.need_exact) {
        // An exact tie at the minimum (need_exact 2) or a window overflowing with near-ties (1: duplicated / quantised inputs).  Which of
        // several exactly tied pairs the reference merges is decided by its heap (ahc_reforder.h), so the problem is recomputed in
        // reference order by the caller.  (Round 2 continued with exact rows and its own tie order here: same heights and partitions on
        // duplicates, but a different row order — and, where tied pairs overlap, possibly a different tree.)
        ++p.fallback;
        p.needs_ro = true;
        p.active = false;
        return FA_SUCCESS;
    } else if (h.halt) { p.active = false; return p.st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: halted without a reason"); }
    return FA_SUCCESS;
}

fa_status prob_finish(fa_ctx *ctx, Prob &p) {   // heights from the stored centroids, dendrogram to the caller's device buffer
    if (p.st != FA_SUCCESS) return p.st;
    if (p.needs_ro) return FA_SUCCESS;   // recomputed by ro_run_device
    if (!p.h.done) return p.st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: round budget exhausted at step %d", p.h.step);
    int32_t hflag = 0;
    hipLaunchKernelGGL(ahc_heights, dim3((p.N + 255) / 256), dim3(256), 0, ctx->stream, p.w);
    FA_HIP_TRY(ctx, hipMemcpyAsync(p.d_Z, p.w.Z, sizeof(double) * 4 * (p.N - 1), p.z_on_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    FA_HIP_TRY(ctx, hipMemcpyAsync(&hflag, p.w.flags, sizeof(hflag), hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (hflag) return p.st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance");
    return FA_SUCCESS;
}

}  // namespace fa_ahc

namespace fa_ahc {
void cached_graph_free(void *p) { delete static_cast<CachedGraph *>(p); }
fa_status ctx_events(fa_ctx *ctx, hipEvent_t (&ev)[3]) {
    for (int i = 0; i < 3; ++i) {
        if (!ctx->ahc_ev[i]) FA_HIP_TRY(ctx, hipEventCreate(&ctx->ahc_ev[i]));
        ev[i] = ctx->ahc_ev[i];
    }
    return FA_SUCCESS;
}
}  // namespace fa_ahc

// ---- adopting a clustering in progress (prob_adopt): what the start-up leaves for the rounds, from a matrix some of whose slots are dead already
namespace {
// Row minimum, lowest-slot argmin and second minimum of every live row over its LIVE columns (the start-up's ahc_row_minima reads a fresh matrix whose dead
// columns hold +inf; here merged-away slots hold whatever their last row was).  One workgroup per row.
__global__ __launch_bounds__(kBlk) void ahc_adopt_rows(Ws w) {
    __shared__ double s_val[kWaves], s_second[kWaves];
    __shared__ int s_idx[kWaves];
    __shared__ int s_win;
    const int i = blockIdx.x;
    double v = dinf(), v2 = dinf();
    int ix = INT_MAX;
    const bool live_row = w.node[i] != kDead;
    if (live_row) {
        const double *row = w.M + static_cast<size_t>(i) * w.Np;
        for (int x = threadIdx.x; x < w.Np; x += kBlk) {
            const double m = (x != i && w.node[x] != kDead) ? row[x] : dinf();
            if (m < v) { v2 = v; v = m; ix = x; }   // x ascending per thread: the lowest index of equal values is kept
            else if (m < v2) v2 = m;
        }
    }
    const double mine = v;
    const int mine_ix = ix;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(ix, off);
        if (lt2(ov, oi, v, ix)) { v = ov; ix = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = v; s_idx[threadIdx.x >> 6] = ix; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kWaves; ++wv) if (lt2(s_val[wv], s_idx[wv], v, ix)) { v = s_val[wv]; ix = s_idx[wv]; }
        RowSt r; r.d1 = v; r.nn = ix == INT_MAX ? -1 : ix; r.nnnode = ix == INT_MAX ? -1 : w.node[ix];
        w.row[i] = r;
        s_win = ix;
    }
    __syncthreads();
    double c2 = mine_ix == s_win ? v2 : mine;   // the smallest entry that is not the winner's
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(c2, off); if (o < c2) c2 = o; }
    if ((threadIdx.x & 63) == 0) s_second[threadIdx.x >> 6] = c2;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kWaves; ++wv) if (s_second[wv] < c2) c2 = s_second[wv];
        w.e2[i] = c2;
    }
}
// the state record after `merges` merges, the window counters, and the dendrogram rows of the merges done (heights are filled by ahc_heights at the end)
__global__ void ahc_adopt_state(Ws w, const int merges, const double eps, const double *__restrict__ pair_a, const double *__restrict__ pair_b) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < merges) {
        const double a = pair_a[r], b = pair_b[r];
        double *z = w.Z + static_cast<size_t>(r) * 4;
        z[0] = a < b ? a : b; z[1] = a < b ? b : a; z[2] = 0.0;
        z[3] = w.sizes[w.N + r];                      // the size of the node this merge created (LinkageOutput::append, FastClusterWrapper.cpp:150-160)
    }
    if (r != 0) return;
    AhcState s{};
    s.mode = FA_AHC_MODE_AUTO;
    s.step = merges;
    for (int k = 0; k < kPend; ++k) { s.pend_row[k] = -1; s.pend_node[k] = -1; }
    s.prev_op = OP_NONE;
    s.n_points = w.N; s.rounds32 = 0;
    s.sym_limit = w.N + merges - 1;                   // every node below the newest has both copies of its pairs; the newest is read through its row (pair_entry)
    s.eps = eps;
    s.dmax_bits = w.state[0].dmax_bits; s.nmax_bits = w.state[0].nmax_bits;
    w.state[0] = s; w.state[1] = s;
    for (int i = 0; i < 4; ++i) { w.cnt[i].stale_key = ~0ULL; w.cnt[i].ncand = 0; w.cnt[i].npairs = 0; }
    w.flags[1] = 0; w.flags[3] = 0;                   // [0]: a NaN distance seen so far stays; [2]: the Gram start-up's
    for (int i = 0; i < 16; ++i) w.prof[i] = 0;
}
}  // namespace

namespace fa_ahc {
fa_status prob_adopt(fa_ctx *ctx, Prob &p, const int merges, const double eps, const double *pair_a, const double *pair_b) {
    const Ws &w = p.w;
    if (p.cpt != 1 || merges < 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ahc: nothing to adopt");
    hipLaunchKernelGGL(ahc_adopt_state, dim3(static_cast<unsigned>((std::max(merges, 1) + 255) / 256)), dim3(256), 0, ctx->stream, w, merges, eps, pair_a, pair_b);
    hipLaunchKernelGGL(ahc_adopt_rows, dim3(w.Np), dim3(kBlk), 0, ctx->stream, w);
    hipLaunchKernelGGL(ahc_records<1>, dim3(w.nblk, 2), dim3(kBlk), 0, ctx->stream, w);   // needs eps (the state) and the rows
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}
}  // namespace fa_ahc

namespace fa_ahc {
fa_status prob_run_rounds(fa_ctx *ctx, Prob &p) {
    const size_t N = p.N, d = p.d, lds = sizeof(double) * d;
    const bool no_single_block = fa::sw_on(fa::Sw::AHC_NO_SINGLE_BLOCK);
    const Ws w = p.w;
    const bool env_big = fa::sw_on(fa::Sw::AHC_ROUND_BIG);
    const bool big = w.nblk > (4 / p.cpt) * 64 || env_big;   // more than 65 536 points (four block records per lane at one slot per thread): the kernel with the many-record reduction
    // records a lane of the first reduction owns: the one-slot-per-thread kernel exists per count (ahc_round_body: a request for a record the lane does not own
    // is not free); the forms with 2 / 4 slots per thread hold all 2 / 1 of theirs
    const int kc = p.cpt == 1 && !big ? std::max(1, (w.nblk + 63) / 64) : 4 / p.cpt;
    auto off_of = [&](const void *p) { return static_cast<unsigned>(static_cast<const char *>(p) - reinterpret_cast<const char *>(w.state)); };   // small arrays: within 4 GB of the state (make_layout puts the matrix last)
    const unsigned o_row = off_of(w.row), o_node = off_of(w.node), o_e2 = off_of(w.e2), o_flags = off_of(w.flags);
    bool lds_ok = true;
    auto launch_as = [&](auto kernel, const int ph, const bool set_lds_only) {
        if (set_lds_only) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) lds_ok = false; return; }
        hipLaunchKernelGGL(kernel, dim3(w.nblk), dim3(kBlk), lds, ctx->stream, ph, w.nblk, w.state, w.recA, w.recI, w.recP, o_row, o_node, o_e2, o_flags, w,
                           static_cast<const Ws *>(nullptr), static_cast<const int2 *>(nullptr));
    };
    auto dispatch = [&](const int ph, const bool set_lds_only) {
        if (p.cpt == 4) { if (big) launch_as(ahc_round_t<false, true, 4>, ph, set_lds_only); else launch_as(ahc_round_t<false, false, 4>, ph, set_lds_only); }
        else if (p.cpt == 2) { if (big) launch_as(ahc_round_t<false, true, 2>, ph, set_lds_only); else launch_as(ahc_round_t<false, false, 2>, ph, set_lds_only); }
        else if (big) launch_as(ahc_round_t<false, true, 1>, ph, set_lds_only);
        else if (kc == 1) launch_as(ahc_round_t<false, false, 1, 1>, ph, set_lds_only);
        else if (kc == 2) launch_as(ahc_round_t<false, false, 1, 2>, ph, set_lds_only);
        else if (kc == 3) launch_as(ahc_round_t<false, false, 1, 3>, ph, set_lds_only);
        else launch_as(ahc_round_t<false, false, 1, 4>, ph, set_lds_only);
    };
    if (lds > 48 * 1024) { dispatch(0, true); (void)lds_ok; (void)hipGetLastError(); }
    auto launch = [&](const int ph) { dispatch(ph, false); };
    // The captured graph only holds launch parameters (workspace pointers, block count): it is reused as long as the workspace sits at
    // the same address and the shape is the same — repeated calls on recordings of one length skip capture + instantiation.
    RoundGraph single_rg;
    RoundGraph *rgp = &single_rg;
    const bool single_block = w.nblk == 1 && !no_single_block;
    if (single_block) {
        single_rg.rounds = rounds_for(N);
        if (lds > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_rounds_single_block<1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_rounds_single_block<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_rounds_single_block<4>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        }
    } else {
        CachedGraph *cg = static_cast<CachedGraph *>(ctx->ahc_graph);
        if (!cg || cg->base != ctx->ahc_ws || cg->N != N || cg->d != d || cg->cpt != p.cpt || !cg->rg.ok) {
            delete cg;
            cg = new CachedGraph();
            ctx->ahc_graph = cg;
            ctx->ahc_graph_free = cached_graph_free;
            cg->base = ctx->ahc_ws; cg->N = N; cg->d = d; cg->cpt = p.cpt;
            cg->rg.capture(ctx, launch, rounds_for(N));
        }
        rgp = &cg->rg;
    }
    RoundGraph &rg = *rgp;
    const long long max_batches = 64 + 8 * static_cast<long long>(N) / rg.rounds;  // bound on rounds (merges + rescans + windows)
    for (long long it = 0; it < max_batches && p.active; ++it) {
        if (single_block) {
            if (p.cpt == 4) hipLaunchKernelGGL(ahc_rounds_single_block<4>, dim3(1), dim3(kBlk), lds, ctx->stream, w, rg.rounds);
            else if (p.cpt == 2) hipLaunchKernelGGL(ahc_rounds_single_block<2>, dim3(1), dim3(kBlk), lds, ctx->stream, w, rg.rounds);
            else hipLaunchKernelGGL(ahc_rounds_single_block<1>, dim3(1), dim3(kBlk), lds, ctx->stream, w, rg.rounds);
            FA_HIP_TRY(ctx, hipGetLastError());
        }
        else FA_TRY(rg.replay(ctx, launch));
        FA_HIP_TRY(ctx, hipMemcpyAsync(&p.h, w.state, sizeof(p.h), hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        FA_TRY(prob_after_replay(ctx, p));
    }
    return prob_finish(ctx, p);
}
}  // namespace fa_ahc

fa_status fa::ahc_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, int mode, fa_ahc_stats *stats, bool z_on_host) {
    // The filter-based rounds keep an N x N matrix resident (N^2 * 8 B); the reference needs O(N d) (fastcluster_internal.hpp:1625-1800).  When the
    // matrix cannot be had — more points than block records (N > 196 608), not enough HBM, or the context's cap — the problem runs in the
    // reference-order mode instead, which has no matrix: slower per merge (every new row is O(N d) exact sums) but the same dendrogram, where
    // round 3 returned ALLOCATION_FAILURE and AHCClustering degraded to singletons (a >= 36 h recording lost its clustering).
    // stats->reference_order == 2 marks that route.
    fa::WsUse ws_use(ctx);                      // released (and trimmed to the context's limit) when the call returns
    auto without_matrix = [&]() {
        if (stats) { *stats = fa_ahc_stats{}; stats->reference_order = 2; }
        const fa_status st = ro_run_device_mf(ctx, d_data, N, d, d_Z, stats, z_on_host);
        if (st == FA_SUCCESS) ctx->last_error.clear();
        return st;
    };
    if (mode == FA_AHC_MODE_REFERENCE_ORDER) {
        if (stats) *stats = fa_ahc_stats{};
        return ro_run_device(ctx, d_data, N, d, d_Z, stats, z_on_host);
    }
    const bool may_fall_back = !fa::sw_on(fa::Sw::AHC_NO_MATRIX_FREE);
    if (prob_check_shape(ctx, N, d) != FA_SUCCESS) {   // too many points for the block records (a too large d fails in ro_run_device as well)
        if (may_fall_back) return without_matrix();
        return FA_ALLOCATION_FAILURE;
    }
    Prob p;
    p.z_on_host = z_on_host;
    // slots per thread of the round: 1 for a chain of its own (the fewest dependent instructions per round: 5.09 us at 43 200 points against 5.60 / 6.69 with
    // 2 / 4); 2 where that makes the problem ONE block (257 .. 512 points: all rounds of a replay inside one launch, no kernel boundary between them:
    // 400 points 2.35 -> 2.08 ms per call; four slots per thread for <= 1 024 points lose to the multi-block chain, 5.0 against 4.9 ms at 900).
    // FA_AHC_CPT forces a value (measurements: profiles/r05_cpt_probe_v2.json).
    const int env_cpt = [] { const char *e = fa::sw(fa::Sw::AHC_CPT); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? v : 0; }();   // per call, like the other switches (the tests flip them)
    const bool no_single_block = fa::sw_on(fa::Sw::AHC_NO_SINGLE_BLOCK);
    p.cpt = env_cpt ? env_cpt : (no_single_block || N <= kBlk || N > 2 * kBlk ? 1 : 2);
    const size_t cols = static_cast<size_t>(kBlk) * p.cpt;
    p.N = N; p.d = d; p.Np = (N + cols - 1) / cols * cols; p.d_data = d_data; p.d_Z = d_Z; p.mode = mode;
    p.L = make_layout(N, p.Np, d, p.Np / cols);
    {
        const fa_status ws = fa::ws_acquire(ctx, p.L.total);
        if (ws == FA_ALLOCATION_FAILURE && may_fall_back) return without_matrix();
        FA_TRY(ws);
    }
    hipEvent_t ev[3];
    FA_TRY(ctx_events(ctx, ev));                // created once per context
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    FA_TRY(prob_setup(ctx, p, static_cast<char *>(ctx->ahc_ws)));
    FA_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));

    FA_TRY(prob_run_rounds(ctx, p));
    FA_HIP_TRY(ctx, hipEventRecord(ev[2], ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (p.needs_ro) {   // exact ties at the minimum: the whole problem again, in the reference's selection order
        if (stats) {
            float t01 = 0, t12 = 0;
            (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
            (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
            *stats = fa_ahc_stats{};
            stats->rounds = p.h.rounds; stats->rescans = p.h.rescans; stats->exact_fallback = p.fallback; stats->windows = p.h.windows;
            stats->init_ms = t01; stats->merge_ms = t12; stats->total_ms = t01 + t12;
        }
        // halted before the first merge (duplicates: the very first window ties), one slot per thread, Gram-form start-up: the matrix in the workspace is the one the
        // reference-order run would build first (ahc_rom.hip shares the arrays)
#ifdef FA_POISON_WORKSPACE
        const bool matrix_ready = false;   // the poisoned build fills the workspace again when the reference-order run acquires it: nothing of the attempt survives
#else
        const bool matrix_ready = mode == FA_AHC_MODE_AUTO && p.h.step == 0 && p.cpt == 1 && d % 16 == 0;
#endif
        return ro_run_device(ctx, d_data, N, d, d_Z, stats, z_on_host, /* may_hand_over = */ mode == FA_AHC_MODE_AUTO, matrix_ready);
    }
#ifdef FA_AHC_PROFILE
    {
        unsigned long long hp[16];
        const Ws &w = p.w;
        (void)hipMemcpy(hp, w.prof, sizeof(hp), hipMemcpyDeviceToHost);
        const double n = hp[15] ? static_cast<double>(hp[15]) : 1.0;
        fprintf(stderr, "ahc profile (cycles/round, block %d of %d, %llu rounds): load+sync %.0f | decide %.0f | merge loads+dab %.0f | row update %.0f | block reduce %.0f | tail %.0f\n",
                w.nblk / 2, w.nblk, hp[15], hp[0] / n, (hp[1] + hp[6] + hp[7] + hp[8]) / n, (hp[2] + hp[9]) / n, hp[3] / n, hp[4] / n, hp[5] / n);
        fprintf(stderr, "  merge loads+dab = operands arrive %.0f | centroid, |ca - cb|^2, wave sum %.0f\n", hp[9] / n, hp[2] / n);
        fprintf(stderr, "  decide = wave reduction %.0f | barrier + result read %.0f | finished rows + global minimum %.0f | state machine + piggy choice %.0f\n", hp[6] / n, hp[7] / n, hp[8] / n, hp[1] / n);
    }
#endif
    if (fa::sw(fa::Sw::AHC_DEBUG))
        fprintf(stderr, "ahc: N %zu rounds %lld merges %d forced re-scans %lld piggy-backed re-scans %lld windows %lld fallback %lld (kPiggy %d)\n", N, p.h.rounds,
                p.h.step, p.h.rescans, p.h.piggy, p.h.windows, p.fallback, kPiggy);
    if (stats) {
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        stats->merges = p.h.step; stats->rounds = p.h.rounds; stats->rescans = p.h.rescans; stats->exact_fallback = p.fallback;
        stats->windows = p.h.windows;
        stats->init_ms = t01; stats->merge_ms = t12; stats->total_ms = t01 + t12;
    }
    return FA_SUCCESS;
}

