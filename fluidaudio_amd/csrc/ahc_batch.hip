// ahc_batch.hip — several linkage problems advanced by the same round launches (ahc_ws.h: the map).
#include "ahc_round_body.h"

using namespace fa_ahc;

namespace {
// ahc_round_uni: K problems in ONE launch without any look-up in front of the round (round 4).  The host lays the K workspaces out with the
// SAME layout (that of the largest problem; a smaller one simply has more dead padding slots) at a constant stride, so every array of
// problem k is the array of problem 0 + k * stride: the grid is (blocks, problems), the problem index is the workgroup id in y (an SGPR the
// hardware hands over), and the addresses of the round's first memory round trip are arithmetic on PRELOADED kernel arguments — the same
// zero scalar round trips as the single-problem kernel.  (ahc_round_args, the round-2 form: 154 scalar instructions and three dependent
// scalar-cache round trips — block -> problem search over 16 block ranges, then two batches of workspace fields out of a by-value array
// indexed by the problem — in front of its first request: 11 us per round of 16 problems against 5.3 us for one.)  Only N differs per
// problem: it comes from the hot part of the problem's state, with the first batch of loads.
// arg 0 = (blocks << 2) | (round & 3), stride in 4 KB pages: 14 preloaded dwords like ahc_round_t.
template <int CPT, int KC = 4 / CPT>
__device__ __forceinline__ void ahc_round_uni_body(const unsigned nblk_ph, const unsigned stride_pages, AhcState *const state_, RecA *const recA_, int4 *const recI_,
                                                   RecP *const recP_, const unsigned off_row, const unsigned off_node, const unsigned off_e2, const unsigned off_flags,
                                                   const Ws &w_one) {
    const size_t sh = (static_cast<size_t>(blockIdx.y) * stride_pages) << 12;
    auto at = [sh](auto *p) { return reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + sh); };
    Ws w_ = w_one;
    const int nblk_ = static_cast<int>(nblk_ph >> 2);
    w_.nblk = nblk_; w_.Np = nblk_ * kBlk * CPT; w_.state = at(state_); w_.recA = at(recA_); w_.recI = at(recI_); w_.recP = at(recP_);
    char *base = reinterpret_cast<char *>(w_.state);
    w_.row = reinterpret_cast<RowSt *>(base + off_row); w_.node = reinterpret_cast<int *>(base + off_node);
    w_.e2 = reinterpret_cast<double *>(base + off_e2); w_.flags = reinterpret_cast<int *>(base + off_flags);
    ahc_round_body<true, false, CPT, KC>(w_, blockIdx.x, static_cast<int>(nblk_ph & 3u), sh);   // the host sends problems of more than 65 536 points elsewhere
}
#define FA_AHC_UNI_KERNEL(NAME, ATTR, CPT, KC)                                                                                                              \
    __global__ __launch_bounds__(kBlk) ATTR void NAME(const unsigned nblk_ph, const unsigned stride_pages, AhcState *const state_, RecA *const recA_,  \
                                                      int4 *const recI_, RecP *const recP_, const unsigned off_row, const unsigned off_node,           \
                                                      const unsigned off_e2, const unsigned off_flags, const Ws w_one) {                              \
        ahc_round_uni_body<CPT, KC>(nblk_ph, stride_pages, state_, recA_, recI_, recP_, off_row, off_node, off_e2, off_flags, w_one);                      \
    }
// one slot per thread at three register budgets (more co-resident workgroups per CU against spills; round 4) and the round-5 forms with 2 / 4 slots per
// thread (which one serves a batch: ahc_batch_uniform)
FA_AHC_UNI_KERNEL(ahc_round_uni, , 1, 4)
FA_AHC_UNI_KERNEL(ahc_round_uni_w3, __attribute__((amdgpu_waves_per_eu(6, 6))), 1, 4)   // "w3" / "w4": the second and third budget
FA_AHC_UNI_KERNEL(ahc_round_uni_w4, __attribute__((amdgpu_waves_per_eu(8, 8))), 1, 4)
FA_AHC_UNI_KERNEL(ahc_round_uni_c2, , 2, 2)
FA_AHC_UNI_KERNEL(ahc_round_uni_c4, , 4, 1)
// the same with the block records a lane of the first reduction actually owns held in registers (round 6: a request for a record the lane does not own is
// not free, ahc_round_body): problems of up to 16 384 / 32 768 / 49 152 slots at one slot per thread, up to 32 768 slots at two
FA_AHC_UNI_KERNEL(ahc_round_uni_k1, , 1, 1)
FA_AHC_UNI_KERNEL(ahc_round_uni_k2, , 1, 2)
FA_AHC_UNI_KERNEL(ahc_round_uni_k3, , 1, 3)
FA_AHC_UNI_KERNEL(ahc_round_uni_c2k1, , 2, 1)

constexpr int kArgProblems = 16;
struct BatchArgs {
    Ws w[kArgProblems];
    int32_t first_block[kArgProblems + 1];   // workgroups [first_block[k], first_block[k + 1]) work on problem k
    int32_t count, pad;
};
static_assert(sizeof(BatchArgs) <= 3584, "kernel arguments are limited to 4 KB");

template <bool BIG>
__global__ __launch_bounds__(kBlk) void ahc_round_args(const BatchArgs a, const int ph) {
    const int b = blockIdx.x;
    int prob = 0;
#pragma unroll
    for (int k = 1; k < kArgProblems; ++k) prob += (k < a.count && b >= a.first_block[k]) ? 1 : 0;
    ahc_round_body<false, BIG>(a.w[prob], b - a.first_block[prob], ph);
}

}  // namespace

// Several independent problems (recordings) advanced by the SAME round launches: one launch = one round of every unfinished
// problem (grid = sum of their blocks), so K serial merge chains share the machine instead of queueing behind each other —
// a chain alone keeps ~N/256 of the 256 CUs busy at one wavefront per SIMD.  Start-up and finish run per problem.
namespace {
fa_status ahc_batch_once(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode,
                         fa_ahc_stats *stats, fa_status *statuses, bool *completed) {
    *completed = false;
    if (mode == FA_AHC_MODE_REFERENCE_ORDER) {   // no batching in this mode: the selection is a serial replay per problem
        fa::WsUse ws_use(ctx);
        fa_status worst = FA_SUCCESS;
        for (int k = 0; k < count; ++k) {
            fa_status st = FA_SUCCESS;
            if (stats) stats[k] = fa_ahc_stats{};
            if (n[k] >= 2) st = ro_run_device(ctx, d_data[k], n[k], d, d_Z[k], stats ? &stats[k] : nullptr);
            if (statuses) statuses[k] = st;
            if (st != FA_SUCCESS && worst == FA_SUCCESS) worst = st;
        }
        *completed = true;
        return worst;
    }
    std::vector<Prob> probs(static_cast<size_t>(count));
    size_t total = 0, total_blocks = 0;
    std::vector<size_t> at(count, 0);
    for (int k = 0; k < count; ++k) {
        Prob &p = probs[k];
        p.N = n[k]; p.d = d; p.Np = (n[k] + kBlk - 1) / kBlk * kBlk; p.d_data = d_data[k]; p.d_Z = d_Z[k]; p.mode = mode;
        if (statuses) statuses[k] = FA_SUCCESS;
        p.st = prob_check_shape(ctx, p.N, d);
        if (p.st != FA_SUCCESS || p.N < 2) { p.active = false; continue; }
        p.L = make_layout(p.N, p.Np, d, p.Np / kBlk);
        at[k] = total;
        total += (p.L.total + 4095) & ~static_cast<size_t>(4095);
        total_blocks += p.Np / kBlk;
    }
    const size_t o_table = total;
    total += (sizeof(Ws) * count + 255) & ~static_cast<size_t>(255);
    const size_t o_map = total;
    total += (sizeof(int2) * std::max<size_t>(total_blocks, 1) + 255) & ~static_cast<size_t>(255);
    fa::WsUse ws_use(ctx);
    FA_TRY(fa::ws_acquire(ctx, total));
    char *base = static_cast<char *>(ctx->ahc_ws);
    hipEvent_t ev[3];
    for (auto &e : ev) FA_HIP_TRY(ctx, hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 3; ++i) (void)hipEventDestroy(e[i]); } } evg{ev};
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    for (int k = 0; k < count; ++k) {
        Prob &p = probs[k];
        if (!p.active) continue;
        const fa_status st = prob_setup(ctx, p, base + at[k]);
        if (st != FA_SUCCESS) { p.st = st; p.active = false; }
    }
    FA_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));
    // table of workspaces + block map of the problems still running (rebuilt only when the set changes a lot: finished problems'
    // workgroups return after one state load, so a stale map is merely idle workgroups)
    std::vector<Ws> table(count);
    for (int k = 0; k < count; ++k) table[k] = probs[k].w;
    const Ws *d_table = reinterpret_cast<const Ws *>(base + o_table);
    int2 *d_map = reinterpret_cast<int2 *>(base + o_map);
    FA_HIP_TRY(ctx, hipMemcpyAsync(const_cast<Ws *>(d_table), table.data(), sizeof(Ws) * count, hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = sizeof(double) * d;
    bool big = fa::sw_on(fa::Sw::AHC_ROUND_BIG);
    for (const Prob &p : probs) if (p.active && p.Np / kBlk > 4 * 64) big = true;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    }
    long long max_batches = 64;
    for (const Prob &p : probs) if (p.active) max_batches = std::max<long long>(max_batches, 64 + 8 * static_cast<long long>(p.N) / rounds_for(p.N));
    std::vector<int2> map;
    int mapped_active = -1;
    RoundGraph *rg = nullptr;
    struct RgGuard { RoundGraph *&p; ~RgGuard() { delete p; } } rgg{rg};
    int grid = 0;
    BatchArgs bargs{};
    bool by_args = false;   // <= kArgProblems running problems: workspaces and block ranges travel in the kernel arguments
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_args<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_args<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    }
    auto launch = [&](const int ph) {
        if (by_args) {
            if (big) hipLaunchKernelGGL(ahc_round_args<true>, dim3(grid), dim3(kBlk), lds, ctx->stream, bargs, ph);
            else hipLaunchKernelGGL(ahc_round_args<false>, dim3(grid), dim3(kBlk), lds, ctx->stream, bargs, ph);
        } else if (big)
            hipLaunchKernelGGL((ahc_round_t<true, true>), dim3(grid), dim3(kBlk), lds, ctx->stream, ph, 0, static_cast<AhcState *>(nullptr), static_cast<RecA *>(nullptr), static_cast<int4 *>(nullptr), static_cast<RecP *>(nullptr), 0u, 0u, 0u, 0u, Ws{}, d_table, static_cast<const int2 *>(d_map));
        else
            hipLaunchKernelGGL((ahc_round_t<true, false>), dim3(grid), dim3(kBlk), lds, ctx->stream, ph, 0, static_cast<AhcState *>(nullptr), static_cast<RecA *>(nullptr), static_cast<int4 *>(nullptr), static_cast<RecP *>(nullptr), 0u, 0u, 0u, 0u, Ws{}, d_table, static_cast<const int2 *>(d_map));
    };
    for (long long it = 0; it < max_batches; ++it) {
        int n_active = 0;
        for (const Prob &p : probs) n_active += p.active ? 1 : 0;
        if (n_active == 0) break;
        if (mapped_active < 0 || n_active * 2 <= mapped_active) {   // (re)build the map and the graph over the running problems
            map.clear();
            for (int k = 0; k < count; ++k)
                if (probs[k].active) for (int b = 0; b < probs[k].w.nblk; ++b) map.push_back(make_int2(k, b));
            grid = static_cast<int>(map.size());
            by_args = n_active <= kArgProblems;
            if (by_args) {
                bargs = BatchArgs{};
                int slot = 0, first = 0;
                for (int k = 0; k < count; ++k)
                    if (probs[k].active) { bargs.w[slot] = probs[k].w; bargs.first_block[slot] = first; first += probs[k].w.nblk; ++slot; }
                bargs.first_block[slot] = first;
                bargs.count = slot;
            }
            FA_HIP_TRY(ctx, hipMemcpyAsync(d_map, map.data(), sizeof(int2) * map.size(), hipMemcpyHostToDevice, ctx->stream));
            FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            delete rg;
            rg = new RoundGraph();
            size_t longest = 0;
            for (const Prob &q : probs) if (q.active && q.N > longest) longest = q.N;
            rg->capture(ctx, launch, rounds_for(longest));
            mapped_active = n_active;
        }
        FA_TRY(rg->replay(ctx, launch));
        for (Prob &p : probs) if (p.active) FA_HIP_TRY(ctx, hipMemcpyAsync(&p.h, p.w.state, sizeof(p.h), hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (Prob &p : probs) if (p.active) (void)prob_after_replay(ctx, p);
    }
    fa_status worst = FA_SUCCESS;
    for (int k = 0; k < count; ++k) {
        Prob &p = probs[k];
        if (p.N >= 2 && p.st == FA_SUCCESS) (void)prob_finish(ctx, p);
        if (statuses) statuses[k] = p.st;
        if (p.st != FA_SUCCESS && worst == FA_SUCCESS) worst = p.st;
    }
    FA_HIP_TRY(ctx, hipEventRecord(ev[2], ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        for (int k = 0; k < count; ++k) {
            const Prob &p = probs[k];
            stats[k] = fa_ahc_stats{};
            stats[k].merges = p.h.step; stats[k].rounds = p.h.rounds; stats[k].rescans = p.h.rescans; stats[k].exact_fallback = p.fallback;
            stats[k].windows = p.h.windows; stats[k].init_ms = t01; stats[k].merge_ms = t12; stats[k].total_ms = t01 + t12;   // times of the whole batch
        }
    }
    // problems that met an exact tie at the minimum: one after the other in reference order (every other problem has delivered its dendrogram)
    for (int k = 0; k < count; ++k) {
        Prob &p = probs[k];
        if (!p.needs_ro || p.st != FA_SUCCESS) continue;
        p.st = ro_run_device(ctx, p.d_data, p.N, p.d, p.d_Z, stats ? &stats[k] : nullptr, false, p.mode == FA_AHC_MODE_AUTO);
        if (statuses) statuses[k] = p.st;
        if (p.st != FA_SUCCESS && worst == FA_SUCCESS) worst = p.st;
    }
    *completed = true;
    return worst;
}
}  // namespace

namespace {
// The same, with the uniform layout of ahc_round_uni: every problem's workspace has the layout of the LARGEST problem and sits at a constant
// stride, the grid is (blocks of that layout, problems).  Eligible batches (the caller checks): >= 2 problems of >= 2 points, no reference-
// order mode, the smallest padded size at least half the largest (a smaller problem only pays dead padding slots: start-up and HBM of
// the larger layout).  Problems are placed by size, largest first: the running set stays a prefix of the placement, so the grid shrinks in y
// as the short ones finish.  Per problem the result is the single-problem entry's bit for bit (test_uniform_batch_*).
int uniform_kernel_choice(size_t blocks_total) {
    // co-residency: 256 CUs x 4 SIMDs x (waves per SIMD) / 4 waves per workgroup.  The round without the many-record path needs 94 VGPRs:
    // 5 waves per SIMD = 1 280 resident workgroups (7 recordings of 8 h).  Capped at 80 / 64 VGPRs (52 / 120 bytes of scratch): 1 536 / 2 048.
    // FA_AHC_UNI_WAVES = 5 | 6 | 8 picks one (measurements: profiles/r04_uni_probe.json).
    (void)blocks_total;
    if (const char *e = fa::sw(fa::Sw::AHC_UNI_WAVES)) { const int v = atoi(e); if (v == 6) return 3; if (v == 8) return 4; }
    return 2;
}

// slots per thread of the round that serves a uniform batch of `count` problems of up to Nmax points (the measurements: ahc_batch_uniform)
int uniform_cpt(int count, size_t Nmax) {
    const size_t wgs1 = static_cast<size_t>(count) * ((Nmax + kBlk - 1) / kBlk);
    return wgs1 >= 450 && Nmax >= 1024 ? 2 : 1;   // (three 8 h recordings: a batch of K = 6 splits into two of three that run side by side: 1 014 workgroups at one slot per thread)
}
// bytes of ONE problem's slot in the uniform layout of such a batch
}  // namespace
namespace fa_ahc {
size_t uniform_stride(int count, size_t Nmax, size_t d) {
    const int cpt = uniform_cpt(count, Nmax);
    const size_t cols = static_cast<size_t>(kBlk) * cpt, Np = (Nmax + cols - 1) / cols * cols;
    return (make_layout(Nmax, Np, d, Np / cols).total + 4095) & ~static_cast<size_t>(4095);
}
}  // namespace fa_ahc
namespace {

fa_status ahc_batch_uniform(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode,
                            fa_ahc_stats *stats, fa_status *statuses, bool *completed) {
    *completed = false;
    std::vector<int> ord(static_cast<size_t>(count));
    for (int k = 0; k < count; ++k) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return n[a] > n[b]; });
    // Slots per thread of the round (ahc_round_body's CPT): a launch over several problems is bound by instruction issue, and a thread that owns four
    // slots leaves a quarter of the workgroups, wavefronts and block records per problem; small problems keep enough blocks to spread over.
    // FA_AHC_UNI_CPT forces 1 / 2 / 4 (measurements).
    const int env_cpt = [] { const char *e = fa::sw(fa::Sw::AHC_UNI_CPT); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? v : 0; }();
    const size_t Nmax = n[ord[0]];
    // Measured (profiles/r05_cpt_probe_v2.json, us per round of 43 200-point problems, one batch): K = 2: 5.80 / 5.93 / 6.83 with 1 / 2 / 4 slots per thread,
    // K = 4: 6.89 / 6.66 / 7.15, K = 8: 10.83 / 7.93 / 8.49, K = 12: 13.41 / 10.00 / 9.80; two batches side by side, K = 8: 8.82 / 7.59 / 8.12, K = 12: 11.17 /
    // 8.18 / 8.90; 16 x 5 400: 6.74 / 6.88 / 7.90.  Two slots per thread pay once a launch holds more than ~2 workgroups per CU at one slot per thread.
    // (Three batches side by side instead of two, profiles/r05_groups_probe.txt: K = 8: 145 -> 152 audio-hours/s linkage-only, K = 12: 187 -> 180: not adopted.)
    const int cpt = env_cpt ? env_cpt : uniform_cpt(count, Nmax);
    const size_t cols = static_cast<size_t>(kBlk) * cpt, Npmax = (Nmax + cols - 1) / cols * cols, nblk = Npmax / cols;
    FA_TRY(prob_check_shape(ctx, Nmax, d));
    const Layout L = make_layout(Nmax, Npmax, d, nblk);
    const size_t stride = (L.total + 4095) & ~static_cast<size_t>(4095);
    if ((stride >> 12) > 0xffffffffull) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: workspace stride too large");
    fa::WsUse ws_use(ctx);
    FA_TRY(fa::ws_acquire(ctx, stride * static_cast<size_t>(count)));
    char *base = static_cast<char *>(ctx->ahc_ws);
    std::vector<Prob> probs(static_cast<size_t>(count));     // in placement order
    hipEvent_t ev[3];
    for (auto &e : ev) FA_HIP_TRY(ctx, hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 3; ++i) (void)hipEventDestroy(e[i]); } } evg{ev};
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    for (int j = 0; j < count; ++j) {
        Prob &p = probs[j];
        const int k = ord[j];
        p.N = n[k]; p.d = d; p.Np = Npmax; p.cpt = cpt; p.d_data = d_data[k]; p.d_Z = d_Z[k]; p.mode = mode; p.L = L;
        if (statuses) statuses[k] = FA_SUCCESS;
        // every problem of the grid gets workgroups, so every state must be initialised: a set-up that fails (a failing launch or copy: the device
        // is in trouble) fails the batch, the caller's splitting logic takes over
        FA_TRY(prob_setup(ctx, p, base + stride * static_cast<size_t>(j)));
    }
    FA_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));
    const size_t lds = sizeof(double) * d;
    const Ws w0 = probs[0].w;
    auto off_of = [&](const void *q) { return static_cast<unsigned>(static_cast<const char *>(q) - reinterpret_cast<const char *>(w0.state)); };
    const unsigned o_row = off_of(w0.row), o_node = off_of(w0.node), o_e2 = off_of(w0.e2), o_flags = off_of(w0.flags);
    const unsigned stride_pages = static_cast<unsigned>(stride >> 12);
    int grid_y = count;
    int kernel = 2;
    auto launch = [&](const int ph) {
        const unsigned a0 = (static_cast<unsigned>(w0.nblk) << 2) | static_cast<unsigned>(ph & 3);
        const dim3 grid(static_cast<unsigned>(w0.nblk), static_cast<unsigned>(grid_y));
        const int lane_recs = (w0.nblk + 63) / 64;   // block records a lane of the first reduction owns
        if (cpt == 4) hipLaunchKernelGGL(ahc_round_uni_c4, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (cpt == 2 && lane_recs == 1) hipLaunchKernelGGL(ahc_round_uni_c2k1, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (cpt == 2) hipLaunchKernelGGL(ahc_round_uni_c2, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (kernel == 2 && lane_recs == 1) hipLaunchKernelGGL(ahc_round_uni_k1, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (kernel == 2 && lane_recs == 2) hipLaunchKernelGGL(ahc_round_uni_k2, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (kernel == 2 && lane_recs == 3) hipLaunchKernelGGL(ahc_round_uni_k3, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (kernel == 4) hipLaunchKernelGGL(ahc_round_uni_w4, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (kernel == 3) hipLaunchKernelGGL(ahc_round_uni_w3, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else hipLaunchKernelGGL(ahc_round_uni, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
    };
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_w3), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_w4), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_c2), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_c4), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_k1), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_k2), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_k3), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_uni_c2k1), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    }
    const long long max_batches = 64 + 8 * static_cast<long long>(Nmax) / rounds_for(Nmax);
    // The captured launches hold the workspace address, the layout (N of the largest problem, d, slots per thread), the grid and the kernel build:
    // the graph of the FIRST capture of a call is kept in the context and reused while all of that is unchanged — a batch job repeats one shape,
    // and capture + instantiation of 512 launches is ~3 ms (6 % of a 16 x 1 h call).  The smaller grids of a shrinking batch are captured per call.
    RoundGraph *rg = nullptr;
    bool rg_owned = false;
    struct RgGuard { RoundGraph *&p; bool &owned; ~RgGuard() { if (owned) delete p; } } rgg{rg, rg_owned};
    int captured_y = -1;
    for (long long it = 0; it < max_batches; ++it) {
        int last_active = -1;
        for (int j = 0; j < count; ++j) if (probs[j].active) last_active = j;
        if (last_active < 0) break;
        // the running set is (nearly) a prefix: shrink the grid when at most half of the captured problems still run
        if (captured_y < 0 || (last_active + 1) * 2 <= captured_y) {
            grid_y = last_active + 1;
            kernel = uniform_kernel_choice(static_cast<size_t>(grid_y) * w0.nblk);
            size_t longest = 0;
            for (int j = 0; j <= last_active; ++j) if (probs[j].active && probs[j].N > longest) longest = probs[j].N;
            if (rg_owned) delete rg;
            rg = nullptr; rg_owned = false;
            const int want_rounds = rounds_for(longest);
            if (grid_y == count) {   // the full grid of the call: the context's cached graph serves it when nothing it bakes in has changed
                CachedGraph *cg = static_cast<CachedGraph *>(ctx->ahc_uni_graph);
                if (!cg || cg->base != base || cg->N != Nmax || cg->d != d || cg->cpt != cpt || cg->grid_y != grid_y || cg->kernel != kernel || cg->rg.rounds != want_rounds || !cg->rg.ok) {
                    delete cg;
                    cg = new CachedGraph();
                    ctx->ahc_uni_graph = cg;
                    ctx->ahc_graph_free = cached_graph_free;
                    cg->base = base; cg->N = Nmax; cg->d = d; cg->cpt = cpt; cg->grid_y = grid_y; cg->kernel = kernel;
                    cg->rg.capture(ctx, launch, want_rounds);
                }
                rg = &cg->rg;
            } else {
                rg = new RoundGraph();
                rg_owned = true;
                rg->capture(ctx, launch, want_rounds);
            }
            captured_y = grid_y;
        }
        FA_TRY(rg->replay(ctx, launch));
        for (int j = 0; j < grid_y; ++j) if (probs[j].active) FA_HIP_TRY(ctx, hipMemcpyAsync(&probs[j].h, probs[j].w.state, sizeof(AhcState), hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int j = 0; j < grid_y; ++j) if (probs[j].active) (void)prob_after_replay(ctx, probs[j]);
    }
    fa_status worst = FA_SUCCESS;
    for (int j = 0; j < count; ++j) {
        Prob &p = probs[j];
        if (p.st == FA_SUCCESS) (void)prob_finish(ctx, p);
        if (statuses) statuses[ord[j]] = p.st;
        if (p.st != FA_SUCCESS && worst == FA_SUCCESS) worst = p.st;
    }
    FA_HIP_TRY(ctx, hipEventRecord(ev[2], ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        for (int j = 0; j < count; ++j) {
            const Prob &p = probs[j];
            fa_ahc_stats &o = stats[ord[j]];
            o = fa_ahc_stats{};
            o.merges = p.h.step; o.rounds = p.h.rounds; o.rescans = p.h.rescans; o.exact_fallback = p.fallback;
            o.windows = p.h.windows; o.init_ms = t01; o.merge_ms = t12; o.total_ms = t01 + t12;   // times of the whole batch
        }
    }
    for (int j = 0; j < count; ++j) {   // exact ties at the minimum: those problems again, one after the other, in reference order
        Prob &p = probs[j];
        if (!p.needs_ro || p.st != FA_SUCCESS) continue;
        p.st = ro_run_device(ctx, p.d_data, p.N, p.d, p.d_Z, stats ? &stats[ord[j]] : nullptr, false, p.mode == FA_AHC_MODE_AUTO);
        if (statuses) statuses[ord[j]] = p.st;
        if (p.st != FA_SUCCESS && worst == FA_SUCCESS) worst = p.st;
    }
    *completed = true;
    return worst;
}

}  // namespace
namespace fa_ahc {
bool uniform_eligible(int count, const size_t *n, int mode) {
    if (count < 2 || mode == FA_AHC_MODE_REFERENCE_ORDER || fa::sw(fa::Sw::AHC_NO_UNIFORM)) return false;
    size_t lo = SIZE_MAX, hi = 0;
    for (int k = 0; k < count; ++k) {
        if (n[k] < 2) return false;
        const size_t np = (n[k] + kBlk - 1) / kBlk * kBlk;
        lo = std::min(lo, np); hi = std::max(hi, np);
    }
    return hi / kBlk >= 2 && lo * 2 >= hi && hi / kBlk <= 4 * 64;   // one-block problems keep their single-launch form; > 65 536 points: the many-record kernels
}
}  // namespace fa_ahc
namespace {
}  // namespace

// Status contract: statuses[k] is the outcome of problem k whatever happens.  An early failure of the batch as a whole (workspace
// allocation, an event, a copy, a graph replay) marks EVERY problem that was to run with that failure — round 2 left them at SUCCESS
// and the callers went on to cut dendrograms that were never written.  When the combined workspace of the batch (sum of N_k^2 * 8 B)
// does not fit, the batch is split in halves down to single problems before anything is reported as ALLOCATION_FAILURE.
namespace {
// A few LARGE problems: their merge chains run CONCURRENTLY, problem 0 on the caller's context and every other one on a helper context
// (own stream, own workspace, a host thread each) — not as one batched chain.  The chain of a large problem is latency-bound (N - 1
// dependent launches on ~N/256 of the 256 CUs, one wavefront per SIMD), so independent chains overlap almost freely: two recordings of
// 43 200 embeddings take 0.29 s this way against 0.36 s as one batched chain and 0.50 s one after the other; four take 0.37 s (one
// hardware queue each: GPU_MAX_HW_QUEUES >= 8 in the process environment helps, profiles/r03_e2e_in_flight.json).  Many SMALL problems
// are the opposite case (a chain of a 5 400-point problem occupies 22 CUs): those stay batched.
constexpr int kInFlightMax = 4;
constexpr size_t kInFlightMinN = 16384;
fa_status ahc_batch_in_flight(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode,
                              fa_ahc_stats *stats, fa_status *sts) {
    for (int k = 1; k < count; ++k) {
        fa_ctx *&h = ctx->helpers[k - 1];
        if (!h) {
            const fa_status st = fa_ctx_create(ctx->device, nullptr, &h);
            if (st != FA_SUCCESS) { h = nullptr; return fa::set_error(ctx, st, "ahc: cannot create a helper context"); }
            h->ws_limit = ctx->ws_limit;
            h->ws_cap = ctx->ws_cap;
        }
    }
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the inputs were produced on the caller's stream
    std::vector<std::thread> threads;
    std::vector<char> started(static_cast<size_t>(count), 0);
    for (int k = 1; k < count; ++k) {
        try {
            threads.emplace_back([&, k]() {
                fa_ctx *h = ctx->helpers[k - 1];
                try {                            // nothing may leave a host thread; ALLOCATION_FAILURE sends the problem to the caller's context below
                    fa::DeviceGuard guard(h->device);
                    sts[k] = fa::ahc_run_device(h, d_data[k], n[k], d, d_Z[k], mode, stats ? &stats[k] : nullptr, false);
                } catch (...) { sts[k] = FA_ALLOCATION_FAILURE; }
            });
            started[static_cast<size_t>(k)] = 1;
        } catch (...) {                          // no thread to be had (std::system_error): that problem runs on the caller's context below
            sts[k] = FA_ALLOCATION_FAILURE;
        }
    }
    sts[0] = fa::ahc_run_device(ctx, d_data[0], n[0], d, d_Z[0], mode, stats ? &stats[0] : nullptr, false);
    for (auto &t : threads) t.join();
    fa_status first = sts[0];
    for (int k = 1; k < count; ++k) {
        if (sts[k] == FA_ALLOCATION_FAILURE) {   // HBM pressure (or no thread): this one runs alone on the caller's context (whose workspace is free again)
            (void)fa_ctx_trim(ctx->helpers[k - 1]);
            sts[k] = fa::ahc_run_device(ctx, d_data[k], n[k], d, d_Z[k], mode, stats ? &stats[k] : nullptr, false);
        }
        if (sts[k] != FA_SUCCESS && ctx->last_error.empty()) ctx->last_error = ctx->helpers[k - 1]->last_error;
        if (first == FA_SUCCESS) first = sts[k];
    }
    return first;
}
}  // namespace

namespace {
// Many LARGE recordings: G uniform batches side by side (round 4).  A uniform batch costs a fixed ~5.3 us per round (kernel boundary + two dependent
// memory round trips: latency) plus ~0.75 us of instruction issue per problem; two batches of K / 2 problems on two streams fill each other's
// latency: 8 recordings of 8 h advance in ~7.5 us per round of both instead of 11 us as one batch.  Group 0 runs on the caller's context, the
// others on its helper contexts (own stream, own workspace, a host thread each — the round-3 in-flight machinery, but with 2 streams instead of
// one per recording, so that two free hardware queues suffice).  A group that cannot get its workspace (or its thread) is run afterwards on the
// caller's context.  FA_AHC_UNI_GROUPS = 1 .. 4 overrides the choice (1: one batch).
fa_status run_device_batch_impl(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode, fa_ahc_stats *stats,
                                fa_status *statuses, bool allow_groups);

constexpr size_t kUniGroupsMinN = 4096;
}  // namespace
namespace fa_ahc {
int uniform_groups(int count, const size_t *n) {
    if (const char *e = fa::sw(fa::Sw::AHC_UNI_GROUPS)) { const int v = atoi(e); if (v >= 1 && v <= 4) return std::min(v, count / 2 > 0 ? count / 2 : 1); }
    size_t lo = SIZE_MAX;
    for (int k = 0; k < count; ++k) lo = std::min(lo, n[k]);
    // six long recordings, or eight medium ones (chains of >= 4 096 rounds: the second stream's thread + graph capture, ~2 ms, must be worth it)
    return (count >= 6 && lo >= kInFlightMinN) || (count >= 8 && lo >= kUniGroupsMinN) ? 2 : 1;
}
}  // namespace fa_ahc
namespace {

fa_status ahc_batch_uniform_groups(fa_ctx *ctx, int groups, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode,
                                   fa_ahc_stats *stats, fa_status *sts) {
    for (int g = 1; g < groups; ++g) {
        fa_ctx *&h = ctx->helpers[g - 1];
        if (!h) {
            if (fa_ctx_create(ctx->device, nullptr, &h) != FA_SUCCESS) { h = nullptr; groups = g; break; }
            h->ws_limit = ctx->ws_limit;
            h->ws_cap = ctx->ws_cap;
        }
    }
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the inputs were produced on the caller's stream
    std::vector<int> first(static_cast<size_t>(groups) + 1, 0);
    for (int g = 0; g <= groups; ++g) first[g] = static_cast<int>(static_cast<long long>(count) * g / groups);
    std::vector<char> done(static_cast<size_t>(groups), 0);
    auto run_group = [&](fa_ctx *c, const int g) {
        const int a = first[g], m = first[g + 1] - first[g];
        bool completed = false;
        try {                                    // nothing may leave a host thread: an exception there would end the process
            fa::DeviceGuard guard(c->device);
            (void)ahc_batch_uniform(c, m, d_data + a, n + a, d, d_Z + a, mode, stats ? stats + a : nullptr, sts + a, &completed);
        } catch (...) { completed = false; }
        done[static_cast<size_t>(g)] = completed ? 1 : 0;
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < groups; ++g) {
        try { threads.emplace_back(run_group, ctx->helpers[g - 1], g); }
        catch (...) { done[static_cast<size_t>(g)] = 0; }   // no thread to be had: that group runs on the caller's context below
    }
    run_group(ctx, 0);
    for (auto &t : threads) t.join();
    fa_status worst = FA_SUCCESS;
    for (int g = 0; g < groups; ++g) {
        const int a = first[g], m = first[g + 1] - first[g];
        if (!done[static_cast<size_t>(g)]) {               // workspace / thread trouble: alone on the caller's context, through the general dispatcher (it splits further)
            if (g > 0 && ctx->helpers[g - 1]) (void)fa_ctx_trim(ctx->helpers[g - 1]);
            (void)run_device_batch_impl(ctx, m, d_data + a, n + a, d, d_Z + a, mode, stats ? stats + a : nullptr, sts + a, false);
        } else if (g > 0 && ctx->last_error.empty()) {
            for (int k = a; k < a + m; ++k) if (sts[k] != FA_SUCCESS) { ctx->last_error = ctx->helpers[g - 1]->last_error; break; }
        }
        for (int k = a; k < a + m; ++k) if (sts[k] != FA_SUCCESS && worst == FA_SUCCESS) worst = sts[k];
    }
    return worst;
}

fa_status run_device_batch_impl(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode, fa_ahc_stats *stats,
                                fa_status *statuses, const bool allow_groups) {
    if (count <= 0) return FA_SUCCESS;
    std::vector<fa_status> local(static_cast<size_t>(count), FA_SUCCESS);
    fa_status *sts = statuses ? statuses : local.data();
    if (count > 1 && mode != FA_AHC_MODE_REFERENCE_ORDER) {
        // A problem the matrix-based rounds cannot hold (more points than block records: N > 196 608) runs alone through the single-problem entry,
        // which takes the matrix-free route (fluidaudio_hip.h promises that; inside a batch such a problem used to be marked ALLOCATION_FAILURE and the
        // clustering stage degraded its recording to singletons).  The others stay a batch.
        std::vector<int> small;
        bool any_big = false;
        for (int k = 0; k < count; ++k) {
            const bool fits = (n[k] + kBlk - 1) / kBlk <= static_cast<size_t>(kMaxBlocks);
            if (fits || n[k] < 2) small.push_back(k); else any_big = true;
        }
        if (any_big) {
            fa_status worst = FA_SUCCESS;
            for (int k = 0; k < count; ++k) {
                if ((n[k] + kBlk - 1) / kBlk <= static_cast<size_t>(kMaxBlocks) || n[k] < 2) continue;
                sts[k] = fa::ahc_run_device(ctx, d_data[k], n[k], d, d_Z[k], mode, stats ? &stats[k] : nullptr, false);
                if (sts[k] != FA_SUCCESS && worst == FA_SUCCESS) worst = sts[k];
            }
            if (!small.empty()) {
                const int m = static_cast<int>(small.size());
                std::vector<const double *> dd(m);
                std::vector<size_t> nn(m);
                std::vector<double *> zz(m);
                std::vector<fa_ahc_stats> ss(m);
                std::vector<fa_status> st2(m, FA_SUCCESS);
                for (int j = 0; j < m; ++j) { dd[j] = d_data[small[j]]; nn[j] = n[small[j]]; zz[j] = d_Z[small[j]]; }
                const fa_status r = run_device_batch_impl(ctx, m, dd.data(), nn.data(), d, zz.data(), mode, stats ? ss.data() : nullptr, st2.data(), allow_groups);
                for (int j = 0; j < m; ++j) { sts[small[j]] = st2[j]; if (stats) stats[small[j]] = ss[j]; }
                if (r != FA_SUCCESS && worst == FA_SUCCESS) worst = r;
            }
            return worst;
        }
    }
    if (allow_groups && uniform_eligible(count, n, mode) && ctx->ws_cap == static_cast<size_t>(-1)) {   // a capped context keeps its promise: ONE workspace within the cap
        const int groups = uniform_groups(count, n);
        if (groups > 1) return ahc_batch_uniform_groups(ctx, groups, count, d_data, n, d, d_Z, mode, stats, sts);
    }
    {
        // chains in flight on helper contexts (round 3) only on request since round 4: the uniform-layout batch advances the same problems by
        // ONE launch per round, on one stream — its rate does not depend on which hardware queues the process's streams landed on
        bool large = count >= 2 && count <= kInFlightMax && fa::sw_on(fa::Sw::AHC_IN_FLIGHT);
        for (int k = 0; k < count && large; ++k) large = n[k] >= kInFlightMinN;
        if (large) return ahc_batch_in_flight(ctx, count, d_data, n, d, d_Z, mode, stats, sts);
    }
    bool completed = false;
    const fa_status st = uniform_eligible(count, n, mode) ? ahc_batch_uniform(ctx, count, d_data, n, d, d_Z, mode, stats, sts, &completed)
                                                          : ahc_batch_once(ctx, count, d_data, n, d, d_Z, mode, stats, sts, &completed);
    if (completed) return st;
    const fa_status fail = st != FA_SUCCESS ? st : FA_RUNTIME_ERROR;
    if (fail == FA_ALLOCATION_FAILURE && count > 1) {
        const int half = count / 2;
        const fa_status a = run_device_batch_impl(ctx, half, d_data, n, d, d_Z, mode, stats, sts, allow_groups);
        const fa_status b = run_device_batch_impl(ctx, count - half, d_data + half, n + half, d, d_Z + half, mode, stats ? stats + half : nullptr, sts + half, allow_groups);
        return a != FA_SUCCESS ? a : b;
    }
    if (fail == FA_ALLOCATION_FAILURE && count == 1 && n[0] >= 2)   // not even one matrix fits: the single-problem entry knows the matrix-free route
        return sts[0] = fa::ahc_run_device(ctx, d_data[0], n[0], d, d_Z[0], mode, stats ? &stats[0] : nullptr, false);
    for (int k = 0; k < count; ++k) {
        if (n[k] >= 2) sts[k] = fail;
        if (stats) stats[k] = fa_ahc_stats{};
    }
    return fail;
}
}  // namespace

fa_status fa::ahc_run_device_batch(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode,
                                   fa_ahc_stats *stats, fa_status *statuses) {
    return run_device_batch_impl(ctx, count, d_data, n, d, d_Z, mode, stats, statuses, true);
}

