// ahc_ro.hip — the reference's selection order, matrix-free (ahc_ws.h: the map; ahc_reforder.h: the selection).
#include "ahc_ws.h"

using namespace fa_ahc;

namespace {
// ------------------------------------------------------------------------------ reference order (ahc_reforder.h)
// The run that reproduces the reference's choice among EXACTLY tied distances: every distance the reference evaluates is evaluated
// here (its sequential fp64 sums), in parallel over the active clusters, and ONE thread replays its selection (binary heap, active
// list, fa_ro::Sel).  Per dendrogram row: ro_scan (all workgroups: the new node against every active node, or the re-scan of a heap
// top whose neighbour is gone; block minima by (value, node id)) + ro_select (one workgroup: the minimum of the block minima, then
// the heap / list updates and the next pair).  O(N d) per merge and two dependent launches: ~40 us per merge instead of 6 — the price
// of the reference's order, paid only by inputs that contain exact ties at the minimum.
__global__ void ro_init(RoWs w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * w.N) { w.sizes[i] = 1.0; w.slot_of[i] = i < w.N ? i : -1; }
    if (i < w.Np) w.node[i] = i < w.N ? i : kDead;
}

// start-up of the reference (fastcluster_internal.hpp:1653-1678): nearest LOWER-indexed point of every point, lowest index on ties — straight from
// the points, no matrix (round 4): the reference itself keeps centroids + nearest-neighbour arrays only (:1625-1800), so this mode runs in O(N d)
// memory like it does, for any N.  Tiles of 64 x 64 pairs on and below the diagonal, 4 x 4 per thread, operands k-major in LDS, every distance =
// the reference's sequential sum (k ascending, one rounding per operation: FastClusterWrapper.cpp:45-52) — the bits ahc_pairwise writes.
constexpr int kRoT = 64, kRoK = 16;
__global__ __launch_bounds__(256) void ro_lower_minima_direct(RoWs w) {
    __shared__ double sa[kRoK][kRoT + 1], sb[kRoK][kRoT + 1];
    const double *__restrict__ x = w.C;          // rows 0 .. N-1 of the centroid store = the input points, row-major
    const int n = w.N, d = w.d;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // tx: column quad, ty: row quad
    const int i0 = blockIdx.x * kRoT;
    double best[4];
    int arg[4];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) { best[r] = dinf(); arg[r] = INT_MAX; }
    for (int j0 = 0; j0 <= i0 && j0 < n; j0 += kRoT) {
        double acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
        for (int k0 = 0; k0 < d; k0 += kRoK) {
            for (int e = tid; e < kRoT * kRoK; e += 256) {
                const int rr = e / kRoK, kk = e % kRoK;
                const int gi = i0 + rr, gj = j0 + rr, gk = k0 + kk;
                sa[kk][rr] = gi < n && gk < d ? x[static_cast<size_t>(gi) * d + gk] : 0.0;
                sb[kk][rr] = gj < n && gk < d ? x[static_cast<size_t>(gj) * d + gk] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < kRoK; ++kk) {
                double av[4], bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { av[r] = sa[kk][4 * ty + r]; bv[r] = sb[kk][4 * tx + r]; }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double diff = __dsub_rn(av[r], bv[c]);
                        acc[r][c] = __dadd_rn(acc[r][c], __dmul_rn(diff, diff));
                    }
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gi = i0 + 4 * ty + r, gj = j0 + 4 * tx + c;
                if (gi < n && gj < gi) {
                    const double v = acc[r][c];
                    if (v != v) bad = true;
                    else if (lt2(v, gj, best[r], arg[r])) { best[r] = v; arg[r] = gj; }
                }
            }
    }
    if (bad) w.flags[0] = 1;                     // NaN distance (nan_error, FastClusterWrapper.cpp:60-62)
#pragma unroll
    for (int r = 0; r < 4; ++r) {                // the 16 threads of a row quad (consecutive lanes): lowest value, then lowest index
        double v = best[r];
        int a = arg[r];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, 16);
            const int oa = __shfl_xor(a, off, 16);
            if (lt2(ov, oa, v, a)) { v = ov; a = oa; }
        }
        const int gi = i0 + 4 * ty + r;
        if (tx == 0 && gi >= 1 && gi < n) { w.key[gi] = v; w.nghbr[gi] = a; }
    }
}

__global__ __launch_bounds__(kBlk) void ro_scan(RoWs w) {
    extern __shared__ double s_c[];            // [d] coordinates of the scanned node
    __shared__ double s_val[kWaves];
    __shared__ int s_idx[kWaves];
    const RoDev st = *w.dev;
    if (st.done || st.op == fa_ro::RO_DONE) return;
    const int tid = threadIdx.x, x = blockIdx.x * kBlk + tid, d = w.d, Np = w.Np;
    const bool fresh = st.op == fa_ro::RO_NEW_ROW;
    const int sa = w.slot_of[st.a], sb = fresh ? w.slot_of[st.b] : -1;
    const int created = st.n + st.merges - 1, limit = fresh ? created : st.a;
    if (fresh) {   // merged centroid (FastClusterWrapper.cpp:89-100); every workgroup evaluates it, workgroup 0 stores it by node id
        const double ma = w.sizes[st.a], mb = w.sizes[st.b], den = ma + mb;
        const double *ca = w.C + static_cast<size_t>(st.a) * d, *cb = w.C + static_cast<size_t>(st.b) * d;
        for (int k = tid; k < d; k += kBlk) {
            const double cc = __ddiv_rn(__dadd_rn(__dmul_rn(ca[k], ma), __dmul_rn(cb[k], mb)), den);
            s_c[k] = cc;
            if (blockIdx.x == 0) w.C[static_cast<size_t>(created) * d + k] = cc;
        }
        if (blockIdx.x == 0 && tid == 0) w.sizes[created] = den;
    } else {
        const double *ca = w.C + static_cast<size_t>(st.a) * d;
        for (int k = tid; k < d; k += kBlk) s_c[k] = ca[k];
    }
    __syncthreads();
    const int nx = w.node[x];
    const bool act = nx != kDead && x != sa && x != sb && nx < limit;
    double sum = dinf();
    if (act) {
        const double *col = w.XT + x;
        sum = 0.0;
#pragma unroll 8
        for (int k = 0; k < d; ++k) {
            const double diff = __dsub_rn(col[static_cast<size_t>(k) * Np], s_c[k]);   // sqeuclidean_extended(j, scanned) (:68-75)
            sum = __dadd_rn(sum, __dmul_rn(diff, diff));
        }
        if (sum != sum) w.flags[0] = 1;
    }
    __syncthreads();                           // every column of this block has been read before slot sa is overwritten
    if (fresh && sa / kBlk == static_cast<int>(blockIdx.x)) {
        for (int k = tid; k < d; k += kBlk) w.XT[static_cast<size_t>(k) * Np + sa] = s_c[k];
        if (tid == 0) { w.node[sa] = created; w.slot_of[created] = sa; }
    }
    if (fresh && x == sb) w.node[sb] = kDead;
    double v = act ? sum : dinf();
    int id = act ? nx : INT_MAX;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(id, off);
        if (lt2(ov, oi, v, id)) { v = ov; id = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = id; }
    __syncthreads();
    if (tid == 0) {
        for (int wv = 1; wv < kWaves; ++wv) if (lt2(s_val[wv], s_idx[wv], v, id)) { v = s_val[wv]; id = s_idx[wv]; }
        RoPart pt; pt.v = v; pt.node = id; pt.pad = 0;
        w.part[blockIdx.x] = pt;
    }
}

__global__ __launch_bounds__(kBlk) void ro_select(RoWs w) {
    __shared__ double s_val[kWaves];
    __shared__ int s_idx[kWaves];
    RoDev st = *w.dev;
    if (st.done || st.op == fa_ro::RO_DONE) return;
    const int tid = threadIdx.x;
    double v = dinf();
    int id = INT_MAX;
    for (int b = tid; b < w.nblk; b += kBlk) { const RoPart pt = w.part[b]; if (lt2(pt.v, pt.node, v, id)) { v = pt.v; id = pt.node; } }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(id, off);
        if (lt2(ov, oi, v, id)) { v = ov; id = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = id; }
    __syncthreads();
    if (tid != 0) return;
    for (int wv = 1; wv < kWaves; ++wv) if (lt2(s_val[wv], s_idx[wv], v, id)) { v = s_val[wv]; id = s_idx[wv]; }
    if (w.flags[0] || id == INT_MAX) { st.done = 1; st.nan_seen = w.flags[0] ? 1 : 2; *w.dev = st; return; }   // NaN distance (nan_error) / nothing to scan
    fa_ro::Sel sel;
    sel.heap.key = w.key; sel.heap.at = w.at; sel.heap.pos = w.pos; sel.heap.size = st.heap_size;
    sel.list.next = w.next; sel.list.prev = w.prev; sel.list.first = st.list_first;
    sel.nghbr = w.nghbr; sel.n = st.n; sel.merges = st.merges; sel.op = st.op; sel.a = st.a; sel.b = st.b;
    sel.pair_a = w.pair_a; sel.pair_b = w.pair_b; sel.height_sq = w.height_sq;
    sel.scan_result(v, id);
    st.heap_size = sel.heap.size; st.list_first = sel.list.first; st.merges = sel.merges; st.op = sel.op; st.a = sel.a; st.b = sel.b;
    st.scans = st.scans + 1;
    if (sel.op == fa_ro::RO_DONE) st.done = 1;
    *w.dev = st;
}

// dendrogram rows as LinkageOutput::append writes them (FastClusterWrapper.cpp:150-160), heights square-rooted (:128-130)
__global__ void ro_finish(RoWs w) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= w.N - 1) return;
    const double a = w.pair_a[r], b = w.pair_b[r];
    double *z = w.Z + static_cast<size_t>(r) * 4;
    z[0] = a < b ? a : b;
    z[1] = a < b ? b : a;
    z[2] = __dsqrt_rn(w.height_sq[r]);
    z[3] = __dadd_rn(w.sizes[static_cast<int>(a)], w.sizes[static_cast<int>(b)]);
}

}  // namespace

namespace fa_ahc {
// the three kernels the matrix-filtered run (ahc_rom.hip) shares with this one, on its own view of the same arrays
void ro_launch_init(hipStream_t st, const RoWs &w, size_t threads) { hipLaunchKernelGGL(ro_init, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, st, w); }
void ro_launch_lower_minima_direct(hipStream_t st, const RoWs &w) { hipLaunchKernelGGL(ro_lower_minima_direct, dim3(static_cast<unsigned>((w.N + kRoT - 1) / kRoT)), dim3(256), 0, st, w); }
void ro_launch_finish(hipStream_t st, const RoWs &w) { hipLaunchKernelGGL(ro_finish, dim3(static_cast<unsigned>((w.N + 255) / 256)), dim3(256), 0, st, w); }

// The whole problem in the reference's selection order (see the kernels above).  d_data / d_Z: device pointers.
fa_status ro_run_device_mf(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host) {
    // O(N d) memory: points / centroids, their slot-major transpose, the reference's heap and list arrays — no distance matrix, so neither the
    // block-record limit of the filter-based rounds nor HBM bounds N here (the start-up computes the nearest lower neighbours tile-wise)
    if (d * sizeof(double) > 60 * 1024) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: dimension too large for the LDS centroid buffer");
    if (N > static_cast<size_t>(INT32_MAX) / 2 - kBlk) return fa::set_error(ctx, FA_INDEX_OVERFLOW, "ahc: N too large for 32-bit node ids");
    const size_t Np = (N + kBlk - 1) / kBlk * kBlk, nblk = Np / kBlk;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~static_cast<size_t>(255); return at; };
    const size_t o_dev = take(sizeof(RoDev)), o_flags = take(16), o_part = take(sizeof(RoPart) * nblk);
    const size_t o_node = take(4 * Np), o_slot = take(4 * 2 * N), o_sizes = take(8 * 2 * N), o_key = take(8 * 2 * N), o_at = take(4 * N), o_pos = take(4 * 2 * N);
    const size_t o_ngh = take(4 * 2 * N), o_next = take(4 * (2 * N + 1)), o_prev = take(4 * (2 * N + 1));
    const size_t o_pa = take(8 * N), o_pb = take(8 * N), o_hs = take(8 * N), o_z = take(8 * 4 * N);
    const size_t o_c = take(8 * d * 2 * N), o_xt = take(8 * d * Np);
    FA_TRY(fa::ws_acquire(ctx, o));
    char *base = static_cast<char *>(ctx->ahc_ws);
    RoWs w{};
    w.dev = reinterpret_cast<RoDev *>(base + o_dev); w.flags = reinterpret_cast<int32_t *>(base + o_flags); w.part = reinterpret_cast<RoPart *>(base + o_part);
    w.node = reinterpret_cast<int32_t *>(base + o_node); w.slot_of = reinterpret_cast<int32_t *>(base + o_slot); w.sizes = reinterpret_cast<double *>(base + o_sizes);
    w.key = reinterpret_cast<double *>(base + o_key); w.at = reinterpret_cast<int32_t *>(base + o_at); w.pos = reinterpret_cast<int32_t *>(base + o_pos);
    w.nghbr = reinterpret_cast<int32_t *>(base + o_ngh); w.next = reinterpret_cast<int32_t *>(base + o_next); w.prev = reinterpret_cast<int32_t *>(base + o_prev);
    w.pair_a = reinterpret_cast<double *>(base + o_pa); w.pair_b = reinterpret_cast<double *>(base + o_pb); w.height_sq = reinterpret_cast<double *>(base + o_hs);
    w.Z = reinterpret_cast<double *>(base + o_z); w.C = reinterpret_cast<double *>(base + o_c); w.XT = reinterpret_cast<double *>(base + o_xt);
    w.N = static_cast<int32_t>(N); w.Np = static_cast<int32_t>(Np); w.d = static_cast<int32_t>(d); w.nblk = static_cast<int32_t>(nblk);
    hipStream_t st = ctx->stream;
    hipEvent_t ev[3];
    FA_TRY(ctx_events(ctx, ev));                // the context's own three events
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], st));
    // ---- start-up: nearest lower-indexed neighbours (the reference's sums), no matrix
    FA_HIP_TRY(ctx, hipMemsetAsync(base + o_dev, 0, o_part - o_dev, st));            // RoDev, flags
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.C, d_data, sizeof(double) * N * d, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(ro_init, dim3(static_cast<unsigned>((std::max(Np, 2 * N) + 255) / 256)), dim3(256), 0, st, w);
    startup_transpose(st, d_data, w.XT, w.N, w.Np, w.d);
    if (N > 1) hipLaunchKernelGGL(ro_lower_minima_direct, dim3(static_cast<unsigned>((N + kRoT - 1) / kRoT)), dim3(256), 0, st, w);
    FA_HIP_TRY(ctx, hipGetLastError());
    // ---- the heap over points 1 .. N-1, the list, the first pair: host (the selection logic is the same header on both sides)
    std::vector<double> key(2 * N, 0.0), pa(N, 0.0), pb(N, 0.0), hs(N, 0.0);
    std::vector<int32_t> at(N, 0), pos(2 * N, 0), ngh(2 * N, 0), next(2 * N + 1, 0), prev(2 * N + 1, 0);
    int32_t hflag = 0;
    FA_HIP_TRY(ctx, hipMemcpyAsync(key.data(), w.key, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(ngh.data(), w.nghbr, sizeof(int32_t) * N, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(&hflag, w.flags, sizeof(hflag), hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (hflag) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance");
    fa_ro::Sel sel{};
    sel.heap.key = key.data(); sel.heap.at = at.data(); sel.heap.pos = pos.data();
    sel.heap.init_identity(static_cast<int32_t>(N) - 1, 1);
    sel.heap.heapify();
    sel.list.next = next.data(); sel.list.prev = prev.data();
    sel.list.init(2 * static_cast<int32_t>(N) - 1);
    sel.nghbr = ngh.data(); sel.n = static_cast<int32_t>(N); sel.merges = 0; sel.pair_a = pa.data(); sel.pair_b = pb.data(); sel.height_sq = hs.data();
    sel.advance();
    RoDev hd{};
    hd.heap_size = sel.heap.size; hd.list_first = sel.list.first; hd.merges = sel.merges; hd.op = sel.op; hd.a = sel.a; hd.b = sel.b; hd.n = sel.n;
    hd.done = sel.op == fa_ro::RO_DONE ? 1 : 0;
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.key, key.data(), sizeof(double) * 2 * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.at, at.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.pos, pos.data(), sizeof(int32_t) * 2 * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.nghbr, ngh.data(), sizeof(int32_t) * 2 * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.next, next.data(), sizeof(int32_t) * (2 * N + 1), hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.prev, prev.data(), sizeof(int32_t) * (2 * N + 1), hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.pair_a, pa.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.pair_b, pb.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.height_sq, hs.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.dev, &hd, sizeof(hd), hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));   // the vectors above are host temporaries
    FA_HIP_TRY(ctx, hipEventRecord(ev[1], st));
    // ---- one (scan, select) pair per dendrogram row or re-scan, replayed from a graph until the device reports the end
    const size_t lds = sizeof(double) * d;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ro_scan), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    auto launch = [&](const int) {
        hipLaunchKernelGGL(ro_scan, dim3(w.nblk), dim3(kBlk), lds, st, w);
        hipLaunchKernelGGL(ro_select, dim3(1), dim3(kBlk), 0, st, w);
    };
    RoundGraph rg;
    rg.capture(ctx, launch, static_cast<int>(std::min<size_t>(256, (N + 3) & ~static_cast<size_t>(3))));
    const long long max_replays = 16 + 8 * static_cast<long long>(N) / rg.rounds;   // rows + re-scans (a node is re-scanned only when it tops the heap with a merged neighbour)
    for (long long it = 0; it < max_replays && !hd.done; ++it) {
        FA_TRY(rg.replay(ctx, launch));
        FA_HIP_TRY(ctx, hipMemcpyAsync(&hd, w.dev, sizeof(hd), hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    if (hd.nan_seen == 1) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance");
    if (!hd.done || hd.nan_seen || hd.merges != static_cast<int32_t>(N) - 1) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: reference-order run stopped at row %d", hd.merges);
    hipLaunchKernelGGL(ro_finish, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, st, w);
    FA_HIP_TRY(ctx, hipMemcpyAsync(d_Z, w.Z, sizeof(double) * 4 * (N - 1), z_on_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    FA_HIP_TRY(ctx, hipEventRecord(ev[2], st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (stats) {
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        stats->merges = hd.merges; stats->rounds += hd.scans; if (!stats->reference_order) stats->reference_order = 1;
        stats->init_ms += t01; stats->merge_ms += t12; stats->total_ms += t01 + t12;
    }
    return FA_SUCCESS;
}


}  // namespace fa_ahc
