// ahc.hip — centroid-linkage agglomerative clustering on gfx950.
//
// Replaces fastcluster_compute_centroid_linkage
//   (reference: Sources/FastClusterWrapper/FastClusterWrapper.cpp:196-244 driving
//    generic_linkage_vector_alternative<METHOD_VECTOR_CENTROID>, fastcluster_internal.hpp:1625-1800)
// and AHCClustering.cluster (Sources/FluidAudio/Diarizer/Offline/Clustering/AHCClustering.swift:20-210).
//
// What the reference computes: N-1 times, merge the globally closest pair of active centroids,
// where the distance of two centroids is the sequential fp64 sum_k (x_k - y_k)^2
// (FastClusterWrapper.cpp:45-52,68-75) and the merged centroid is (m_i x_i + m_j x_j)/(m_i+m_j)
// (:89-100).  Its heap / nearest-neighbour arrays are bookkeeping for that argmin.
//
// How this file computes the same thing:
//   * vectors live transposed in HBM, XT[k][slot]; a merged cluster keeps the lower slot;
//   * the full slot x slot distance matrix M (fp64, N^2*8 B: 20 GB at N = 50 000) stays
//     resident in HBM; dead rows/columns hold +inf so row scans need no mask;
//   * per row: (rowmin, rownn, valid).  A row whose nearest neighbour was merged away keeps its
//     old minimum as a LOWER BOUND (valid = 0) and is re-scanned only when that bound reaches
//     the global minimum — the reference's lazy scheme, applied to full rows;
//   * one merge = two kernels replayed from a hipGraph: `select` (1 workgroup: finish the
//     previous round's reduction, global argmin over 256-row block minima, exact re-evaluation
//     of the winning pair, new centroid, dendrogram row) and `apply` (N/256 workgroups: new
//     matrix row/column, kill the dead column, maintain row minima and block minima);
//   * FA_AHC_MODE_AUTO fills the new row with the Lance-Williams centroid update of rows a, b
//     (O(N) per merge instead of O(N d)) and re-evaluates the selected pair with the
//     reference's exact sum; if two candidates ever fall within the rounding bound eps of each
//     other, the run recomputes M exactly and continues with exact rows (FA_AHC_MODE_EXACT),
//     so the merge sequence never depends on the approximation.
// Exactly tied distances are merged in (lower slot, higher slot) order, where a cluster's slot
// is its smallest original point index; the reference's tie order is an artefact of its binary
// heap layout.  Heights and the partition at any threshold are the same (tests/test_ahc_*.py).
#include <algorithm>
#include <climits>
#include <cmath>
#include <mutex>
#include <vector>

#include "fa_common.h"

namespace {

constexpr int kBlk = 256;       // rows per block-minimum / threads per apply workgroup
constexpr int kSelThreads = 1024;
constexpr int kMaxBlocks = 2048;  // N <= 524 288 (the N^2 matrix limits N far earlier)
constexpr int kRoundsPerGraph = 512;

constexpr int kMaxCand = 64;     // candidate rows inside an ambiguity window
constexpr int kMaxPairs = 1024;  // matrix entries inside an ambiguity window

enum { OP_NOOP = 0, OP_MERGE = 1, OP_RESCAN = 2, OP_WINDOW = 3 };

struct AhcState {
    int32_t step, done, halt, need_exact, error, op, a, b, r, mode;
    double dab, wa, wb, wab, eps;
    unsigned long long dmax_bits;
    long long rounds, rescans, windows;
    double lim;
    int32_t ncand, npairs;
};

struct Ws {
    double *XT;      // [d][Np]
    double *M;       // [Np][Np]
    double *rowmin;  // [Np]
    double *size;    // [Np]
    double *bm;      // [nblk]
    double *pval;    // [nblk]
    double *cvec;    // [d]
    double *Z;       // [(N-1)*4]
    int32_t *rownn, *valid, *active, *node, *pidx;
    int32_t *cand;   // [kMaxCand]
    int32_t *pairs;  // [2*kMaxPairs]
    AhcState *state;
    int32_t N, Np, d, nblk;
};

__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

// (value, index) minimum, lower index on ties.  Result valid in every thread.
template <int THREADS>
__device__ __forceinline__ void block_argmin(double &v, int &ix, double *s_val, int *s_idx) {
    constexpr int W = THREADS / 64;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(ix, off);
        if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { s_val[wave] = v; s_idx[wave] = ix; }
    __syncthreads();
    v = s_val[0]; ix = s_idx[0];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const double ov = s_val[w];
        const int oi = s_idx[w];
        if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------ init kernels
__global__ void ahc_transpose(const double *__restrict__ data, double *__restrict__ XT, int N, int Np, int d) {
    __shared__ double tile[32][33];
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, k = k0 + tx;
        tile[r][tx] = (i < N && k < d) ? data[static_cast<size_t>(i) * d + k] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, i = i0 + tx;
        if (k < d && i < Np) XT[static_cast<size_t>(k) * Np + i] = tile[tx][r];
    }
}

__global__ void ahc_init_rows(Ws w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.Np) return;
    w.active[i] = i < w.N;
    w.node[i] = i;
    w.size[i] = 1.0;
    w.valid[i] = 1;
    w.rownn[i] = -1;
    w.rowmin[i] = dinf();
}

// Exact pairwise squared distances, the reference's summation order (FastClusterWrapper.cpp:45-52).
constexpr int PT = 64, PK = 16;
__global__ __launch_bounds__(256) void ahc_pairwise(Ws w) {
    __shared__ double sa[PK][PT];
    __shared__ double sb[PK][PT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * PT, j0 = blockIdx.x * PT;
    const int Np = w.Np, d = w.d;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int k0 = 0; k0 < d; k0 += PK) {
        for (int e = tid; e < PK * PT; e += 256) {
            const int kk = e / PT, c = e % PT, k = k0 + kk;
            sa[kk][c] = k < d ? w.XT[static_cast<size_t>(k) * Np + i0 + c] : 0.0;
            sb[kk][c] = k < d ? w.XT[static_cast<size_t>(k) * Np + j0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < PK; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = sa[kk][ty * 4 + r]; bv[r] = sb[kk][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double diff = __dsub_rn(av[r], bv[c]);
                    acc[r][c] = __dadd_rn(acc[r][c], __dmul_rn(diff, diff));  // one rounding per op, k ascending
                }
        }
        __syncthreads();
    }
    double lmax = 0.0;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty * 4 + r;
        const bool ai = i < w.N && w.active[i];
        double out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx * 4 + c;
            const bool ok = ai && j < w.N && i != j && w.active[j];
            const double v = acc[r][c];
            if (ok) { if (v != v) bad = true; else if (v > lmax) lmax = v; }
            out[c] = ok ? v : dinf();
        }
        double *dst = w.M + static_cast<size_t>(i) * Np + j0 + tx * 4;
        reinterpret_cast<double2 *>(dst)[0] = make_double2(out[0], out[1]);
        reinterpret_cast<double2 *>(dst)[1] = make_double2(out[2], out[3]);
    }
    if (bad) w.state->error = 1;  // nan_error (fastcluster_internal.hpp / FastClusterWrapper.cpp:60-62)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(lmax, off); if (o > lmax) lmax = o; }
    if ((tid & 63) == 0 && lmax > 0.0)
        atomicMax(&w.state->dmax_bits, static_cast<unsigned long long>(__double_as_longlong(lmax)));
}

// Row minimum + lowest-index argmin of every active row (one workgroup per row).
__global__ __launch_bounds__(kBlk) void ahc_row_minima(Ws w) {
    __shared__ double s_val[kBlk / 64];
    __shared__ int s_idx[kBlk / 64];
    const int i = blockIdx.x;
    double v = dinf();
    int ix = INT_MAX;
    if (w.active[i]) {
        const double *row = w.M + static_cast<size_t>(i) * w.Np;
        for (int x = threadIdx.x; x < w.Np; x += kBlk) {
            const double m = row[x];
            if (m < v) { v = m; ix = x; }  // x ascending per thread => lowest index kept
        }
    }
    block_argmin<kBlk>(v, ix, s_val, s_idx);
    if (threadIdx.x == 0) {
        w.rowmin[i] = v;
        w.rownn[i] = ix == INT_MAX ? -1 : ix;
        w.valid[i] = 1;
    }
}

__global__ __launch_bounds__(kBlk) void ahc_block_minima(Ws w) {
    __shared__ double s_val[kBlk / 64];
    __shared__ int s_idx[kBlk / 64];
    const int x = blockIdx.x * kBlk + threadIdx.x;
    double v = w.active[x] ? w.rowmin[x] : dinf();
    int ix = x;
    block_argmin<kBlk>(v, ix, s_val, s_idx);
    if (threadIdx.x == 0) w.bm[blockIdx.x] = v;
}

// ------------------------------------------------------------------------------ round: select
// Exact squared distances of up to kMaxPairs slot pairs, the reference's summation order
// (sequential in k, one rounding per operation; FastClusterWrapper.cpp:68-75).  One wavefront per
// pair: 64 lanes square the differences of a 64-wide slice, lane 0 adds them in index order.
// Returns the minimum (ties -> lexicographically lowest (a, b)) in every thread.
__device__ void exact_min_pair(const Ws &w, const int *s_pa, const int *s_pb, const int np, double *s_sq /*[16][64]*/,
                               double *s_val, int *s_idx, double &best, int &best_a, int &best_b) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Np = w.Np, d = w.d;
    best = dinf();
    int best_p = INT_MAX;
    for (int p0 = 0; p0 < np; p0 += kSelThreads / 64) {
        const int p = p0 + wave;
        const bool live = p < np;
        const int a = live ? s_pa[p] : 0, b = live ? s_pb[p] : 0;
        double sum = 0.0;
        for (int k0 = 0; k0 < d; k0 += 64) {
            const int k = k0 + lane;
            double sq = 0.0;
            if (live && k < d) {
                const double diff = __dsub_rn(w.XT[static_cast<size_t>(k) * Np + a], w.XT[static_cast<size_t>(k) * Np + b]);
                sq = __dmul_rn(diff, diff);
            }
            s_sq[wave * 64 + lane] = sq;
            __syncthreads();
            if (lane == 0 && live) {
                const int n = d - k0 < 64 ? d - k0 : 64;
                for (int j = 0; j < n; ++j) sum = __dadd_rn(sum, s_sq[wave * 64 + j]);
            }
            __syncthreads();
        }
        if (lane == 0 && live) {
            // order pairs by (value, a, b): encode (a, b) through the pair's position after the value compare
            if (sum != sum) { best = sum; best_p = -1; }  // NaN: remember it, reported by the caller
            else if (best_p != -1 && (sum < best || (sum == best && (best_p == INT_MAX || a < s_pa[best_p] || (a == s_pa[best_p] && b < s_pb[best_p]))))) {
                best = sum; best_p = p;
            }
        }
    }
    // cross-wave reduction (lane 0 of each wave holds its candidate)
    __syncthreads();
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = best_p; }
    __syncthreads();
    best = dinf(); best_p = INT_MAX;
    for (int wv = 0; wv < kSelThreads / 64; ++wv) {
        const double v = s_val[wv];
        const int p = s_idx[wv];
        if (p == -1) { best = v; best_p = -1; break; }
        if (p == INT_MAX) continue;
        if (best_p == INT_MAX || v < best || (v == best && (s_pa[p] < s_pa[best_p] || (s_pa[p] == s_pa[best_p] && s_pb[p] < s_pb[best_p])))) {
            best = v; best_p = p;
        }
    }
    __syncthreads();
    best_a = best_p >= 0 && best_p != INT_MAX ? s_pa[best_p] : -1;
    best_b = best_p >= 0 && best_p != INT_MAX ? s_pb[best_p] : -1;
}

__global__ __launch_bounds__(kSelThreads) void ahc_select(Ws w) {
    __shared__ AhcState st;
    __shared__ double s_val[kSelThreads / 64];
    __shared__ int s_idx[kSelThreads / 64];
    __shared__ double s_bm[kMaxBlocks];
    __shared__ double s_sq[kSelThreads];
    __shared__ int s_pa[kMaxPairs], s_pb[kMaxPairs];
    __shared__ int s_cand[kMaxCand];
    __shared__ int s_stale, s_best, s_cnt, s_nc;
    const int tid = threadIdx.x;
    AhcState *S = w.state;
    if (tid == 0) st = *S;
    __syncthreads();
    if (st.done || st.halt) { if (tid == 0 && st.op != OP_NOOP) S->op = OP_NOOP; return; }
    const int Np = w.Np, nblk = w.nblk, d = w.d;
    int np = 0;        // pairs to evaluate exactly
    double v = dinf();  // approximate (or, in exact mode, exact) value of the selected pair

    // (1) finish the previous round
    if (st.op == OP_MERGE || st.op == OP_RESCAN) {  // reduce the per-block partial minima of the row it produced
        const int row = st.op == OP_MERGE ? st.a : st.r;
        double rv = dinf();
        int ix = INT_MAX;
        for (int i = tid; i < nblk; i += kSelThreads) {
            const double pv = w.pval[i];
            const int pi = w.pidx[i];
            if (pv < rv || (pv == rv && pi < ix)) { rv = pv; ix = pi; }
        }
        block_argmin<kSelThreads>(rv, ix, s_val, s_idx);
        if (tid == 0) {
            w.rowmin[row] = rv;
            w.rownn[row] = ix == INT_MAX ? -1 : ix;
            w.valid[row] = 1;
        }
        __syncthreads();
        const int blk = row / kBlk;
        double bv = dinf();
        int bi = INT_MAX;
        if (tid < kBlk) { const int x = blk * kBlk + tid; if (w.active[x]) { bv = w.rowmin[x]; bi = x; } }
        block_argmin<kSelThreads>(bv, bi, s_val, s_idx);
        if (tid == 0) w.bm[blk] = bv;
        __syncthreads();
    } else if (st.op == OP_WINDOW) {  // the apply pass listed every pair inside the ambiguity window
        np = st.npairs;
        if (np < 1 || np > kMaxPairs) {  // massive ties (duplicated inputs): continue with exact rows
            if (tid == 0) { S->need_exact = 1; S->halt = 1; S->op = OP_NOOP; }
            return;
        }
        for (int i = tid; i < np; i += kSelThreads) { s_pa[i] = w.pairs[2 * i]; s_pb[i] = w.pairs[2 * i + 1]; }
        __syncthreads();
    }
    if (st.step >= w.N - 1) {
        if (tid == 0) { S->done = 1; S->op = OP_NOOP; }
        return;
    }

    if (np == 0) {
        // (2) global minimum over block minima; candidate rows within 2*eps of it
        for (int i = tid; i < nblk; i += kSelThreads) s_bm[i] = w.bm[i];
        if (tid == 0) { s_stale = INT_MAX; s_best = INT_MAX; s_cnt = 0; s_nc = 0; }
        __syncthreads();
        int vi = INT_MAX;
        for (int i = tid; i < nblk; i += kSelThreads) if (s_bm[i] < v) { v = s_bm[i]; vi = i; }
        block_argmin<kSelThreads>(v, vi, s_val, s_idx);
        const double lim = v + 2.0 * st.eps;
        for (int blk0 = 0; blk0 < nblk; blk0 += kSelThreads / kBlk) {
            const int blk = blk0 + tid / kBlk;
            if (blk < nblk && s_bm[blk] <= lim) {
                const int x = blk * kBlk + (tid & (kBlk - 1));
                const double rm = w.rowmin[x];
                if (w.active[x] && rm <= lim) {
                    if (!w.valid[x]) atomicMin(&s_stale, x);
                    else {
                        atomicAdd(&s_cnt, 1);
                        if (rm == v) atomicMin(&s_best, x);
                        const int i = atomicAdd(&s_nc, 1);
                        if (i < kMaxCand) s_cand[i] = x;
                    }
                }
            }
        }
        __syncthreads();
        if (s_stale != INT_MAX) {  // a lower bound reached the minimum: re-scan that row first
            if (tid == 0) { S->op = OP_RESCAN; S->r = s_stale; S->rescans = st.rescans + 1; S->rounds = st.rounds + 1; }
            return;
        }
        const int r = s_best;
        if (r == INT_MAX || !(v < dinf())) {  // cannot happen with finite data; stop rather than spin
            if (tid == 0) { S->error = 2; S->halt = 1; S->op = OP_NOOP; }
            return;
        }
        const int q = w.rownn[r];
        if (st.mode == FA_AHC_MODE_AUTO) {
            const bool mutual = s_cnt == 2 && q >= 0 && w.valid[q] && w.rownn[q] == r && w.rowmin[q] <= lim;
            if (!mutual) {
                // Several pairs lie within the rounding bound of the Lance-Williams rows.  Ask the apply pass to list
                // every matrix entry <= lim in the candidate rows; the next select evaluates them exactly.
                if (s_nc > kMaxCand) {
                    if (tid == 0) { S->need_exact = 1; S->halt = 1; S->op = OP_NOOP; }
                    return;
                }
                for (int i = tid; i < s_nc; i += kSelThreads) w.cand[i] = s_cand[i];
                if (tid == 0) {
                    S->op = OP_WINDOW; S->ncand = s_nc; S->npairs = 0; S->lim = lim;
                    S->windows = st.windows + 1; S->rounds = st.rounds + 1;
                }
                return;
            }
        }
        if (tid == 0) { s_pa[0] = r < q ? r : q; s_pb[0] = r < q ? q : r; }
        np = 1;
        __syncthreads();
    }

    // (3) the reference's exact distance of the selected pair(s)
    double dab = v;
    int a = s_pa[0], b = s_pb[0];
    if (st.mode == FA_AHC_MODE_AUTO) exact_min_pair(w, s_pa, s_pb, np, s_sq, s_val, s_idx, dab, a, b);
    if (a < 0 || dab != dab) {  // NaN distance: nan_error in the reference (status 5)
        if (tid == 0) { S->error = 1; S->halt = 1; S->op = OP_NOOP; }
        return;
    }
    // (4) merged centroid (FastClusterWrapper.cpp:89-100) into slot a, plus a contiguous copy
    const double ma = w.size[a], mb = w.size[b], den = ma + mb;
    __syncthreads();
    for (int k = tid; k < d; k += kSelThreads) {
        const double xa = w.XT[static_cast<size_t>(k) * Np + a], xb = w.XT[static_cast<size_t>(k) * Np + b];
        const double c = __ddiv_rn(__dadd_rn(__dmul_rn(xa, ma), __dmul_rn(xb, mb)), den);
        w.XT[static_cast<size_t>(k) * Np + a] = c;
        w.cvec[k] = c;
    }
    if (tid == 0) {
        const int na = w.node[a], nb = w.node[b];
        double *z = w.Z + static_cast<size_t>(st.step) * 4;
        z[0] = na < nb ? na : nb;  // LinkageOutput::append (FastClusterWrapper.cpp:150-160)
        z[1] = na < nb ? nb : na;
        z[2] = dab;                // squared; sqrt applied after the loop (postprocess, :128-130)
        z[3] = den;
        w.size[a] = den;
        w.node[a] = w.N + st.step;
        w.active[b] = 0;
        w.rowmin[b] = dinf();
        w.valid[b] = 1;
        S->op = OP_MERGE; S->a = a; S->b = b; S->dab = dab;
        S->wa = ma / den; S->wb = mb / den; S->wab = (ma * mb) / (den * den);
        S->step = st.step + 1;
        S->rounds = st.rounds + 1;
    }
}

// ------------------------------------------------------------------------------ round: apply
__global__ __launch_bounds__(kBlk) void ahc_apply(Ws w) {
    __shared__ double s_val[kBlk / 64];
    __shared__ int s_idx[kBlk / 64];
    const AhcState *S = w.state;
    const int op = S->op;
    if (op == OP_NOOP) return;
    const int tid = threadIdx.x, blk = blockIdx.x, x = blk * kBlk + tid;
    const int Np = w.Np;
    if (op == OP_RESCAN) {
        double v = w.M[static_cast<size_t>(S->r) * Np + x];
        int ix = x;
        block_argmin<kBlk>(v, ix, s_val, s_idx);
        if (tid == 0) { w.pval[blk] = v; w.pidx[blk] = v < dinf() ? ix : INT_MAX; }
        return;
    }
    if (op == OP_WINDOW) {  // list every entry of the candidate rows that lies inside the ambiguity window
        const int nc = S->ncand;
        const double lim = S->lim;
        for (int j = 0; j < nc; ++j) {
            const int row = w.cand[j];
            const double val = w.M[static_cast<size_t>(row) * Np + x];
            if (val <= lim) {
                const int slot = atomicAdd(&w.state->npairs, 1);
                if (slot < kMaxPairs) { w.pairs[2 * slot] = row < x ? row : x; w.pairs[2 * slot + 1] = row < x ? x : row; }
            }
        }
        return;
    }
    const int a = S->a, b = S->b;
    const bool act = w.active[x] != 0 && x != a;
    double dc = dinf();
    if (S->mode == FA_AHC_MODE_AUTO) {
        const double da = w.M[static_cast<size_t>(a) * Np + x];
        const double db = w.M[static_cast<size_t>(b) * Np + x];
        if (act) {  // Lance-Williams centroid update; only a filter, the winner is re-evaluated exactly
            dc = S->wa * da + S->wb * db - S->wab * S->dab;
            if (dc < 0.0) dc = 0.0;
        }
    } else {
        const int d = w.d;
        const double *col = w.XT + x;
        double sum = 0.0;
#pragma unroll 8
        for (int k = 0; k < d; ++k) {
            const double diff = __dsub_rn(w.cvec[k], col[static_cast<size_t>(k) * Np]);
            sum = __dadd_rn(sum, __dmul_rn(diff, diff));  // sqeuclidean_extended (FastClusterWrapper.cpp:68-75)
        }
        if (act) { dc = sum; if (sum != sum) w.state->error = 1; }
    }
    w.M[static_cast<size_t>(a) * Np + x] = dc;
    double rm = dinf();
    if (act) {
        w.M[static_cast<size_t>(x) * Np + a] = dc;
        w.M[static_cast<size_t>(x) * Np + b] = dinf();
        rm = w.rowmin[x];
        const int nn = w.rownn[x];
        const int vld = w.valid[x];
        if (dc < rm || (vld && dc == rm && a <= nn)) {
            rm = dc;
            w.rowmin[x] = dc; w.rownn[x] = a; w.valid[x] = 1;
        } else if (vld && (nn == a || nn == b)) {
            w.valid[x] = 0;  // minimum lost: rm stays as a lower bound
        }
    }
    // block minimum of row minima (row a is finished by the next select)
    double bv = rm;
    int bi = x;
    block_argmin<kBlk>(bv, bi, s_val, s_idx);
    if (tid == 0) w.bm[blk] = bv;
    // partial minimum of the new row
    double pv = dc;
    int pi = x;
    block_argmin<kBlk>(pv, pi, s_val, s_idx);
    if (tid == 0) { w.pval[blk] = pv; w.pidx[blk] = pv < dinf() ? pi : INT_MAX; }
}

__global__ void ahc_finish(Ws w) {  // heights: squared -> Euclidean (cluster_result::sqrt, FastClusterWrapper.cpp:128-130)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < w.N - 1) w.Z[static_cast<size_t>(i) * 4 + 2] = __dsqrt_rn(w.Z[static_cast<size_t>(i) * 4 + 2]);
}

// ------------------------------------------------------------------------------ host driver
struct Layout {
    size_t xt, m, rowmin, size, bm, pval, cvec, z, rownn, valid, active, node, pidx, cand, pairs, state, total;
};

Layout make_layout(size_t N, size_t Np, size_t d, size_t nblk) {
    Layout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~static_cast<size_t>(255); return at; };
    L.state = take(sizeof(AhcState));
    L.xt = take(sizeof(double) * d * Np);
    L.rowmin = take(sizeof(double) * Np);
    L.size = take(sizeof(double) * Np);
    L.bm = take(sizeof(double) * nblk);
    L.pval = take(sizeof(double) * nblk);
    L.cvec = take(sizeof(double) * d);
    L.z = take(sizeof(double) * 4 * (N > 1 ? N - 1 : 1));
    L.rownn = take(sizeof(int32_t) * Np);
    L.valid = take(sizeof(int32_t) * Np);
    L.active = take(sizeof(int32_t) * Np);
    L.node = take(sizeof(int32_t) * Np);
    L.pidx = take(sizeof(int32_t) * nblk);
    L.cand = take(sizeof(int32_t) * kMaxCand);
    L.pairs = take(sizeof(int32_t) * 2 * kMaxPairs);
    L.m = take(sizeof(double) * Np * Np);
    L.total = o;
    return L;
}

fa_status exact_rebuild(fa_ctx *ctx, const Ws &w) {
    const int tiles = w.Np / PT;
    hipLaunchKernelGGL(ahc_pairwise, dim3(tiles, tiles), dim3(256), 0, ctx->stream, w);
    hipLaunchKernelGGL(ahc_row_minima, dim3(w.Np), dim3(kBlk), 0, ctx->stream, w);
    hipLaunchKernelGGL(ahc_block_minima, dim3(w.nblk), dim3(kBlk), 0, ctx->stream, w);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

// d_data: device [N][d]; d_Z: device [(N-1)*4] (heights already square-rooted on return).
fa_status ahc_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, int mode, fa_ahc_stats *stats) {
    const size_t Np = (N + kBlk - 1) / kBlk * kBlk;
    const size_t nblk = Np / kBlk;
    if (nblk > kMaxBlocks) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: N too large for the resident distance matrix");
    const Layout L = make_layout(N, Np, d, nblk);
    if (ctx->ahc_ws_bytes < L.total) {
        if (ctx->ahc_ws) { FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->ahc_ws); ctx->ahc_ws = nullptr; ctx->ahc_ws_bytes = 0; }
        const hipError_t e = hipMalloc(&ctx->ahc_ws, L.total);
        if (e != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: cannot allocate %zu bytes of HBM", L.total); }
        ctx->ahc_ws_bytes = L.total;
    }
    char *base = static_cast<char *>(ctx->ahc_ws);
    Ws w{};
    w.state = reinterpret_cast<AhcState *>(base + L.state);
    w.XT = reinterpret_cast<double *>(base + L.xt);
    w.M = reinterpret_cast<double *>(base + L.m);
    w.rowmin = reinterpret_cast<double *>(base + L.rowmin);
    w.size = reinterpret_cast<double *>(base + L.size);
    w.bm = reinterpret_cast<double *>(base + L.bm);
    w.pval = reinterpret_cast<double *>(base + L.pval);
    w.cvec = reinterpret_cast<double *>(base + L.cvec);
    w.Z = reinterpret_cast<double *>(base + L.z);
    w.rownn = reinterpret_cast<int32_t *>(base + L.rownn);
    w.valid = reinterpret_cast<int32_t *>(base + L.valid);
    w.active = reinterpret_cast<int32_t *>(base + L.active);
    w.node = reinterpret_cast<int32_t *>(base + L.node);
    w.pidx = reinterpret_cast<int32_t *>(base + L.pidx);
    w.cand = reinterpret_cast<int32_t *>(base + L.cand);
    w.pairs = reinterpret_cast<int32_t *>(base + L.pairs);
    w.N = static_cast<int32_t>(N); w.Np = static_cast<int32_t>(Np); w.d = static_cast<int32_t>(d); w.nblk = static_cast<int32_t>(nblk);

    hipEvent_t ev[3];
    for (auto &e : ev) FA_HIP_TRY(ctx, hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 3; ++i) (void)hipEventDestroy(e[i]); } } evg{ev};

    AhcState init{};
    init.mode = mode == FA_AHC_MODE_EXACT ? FA_AHC_MODE_EXACT : FA_AHC_MODE_AUTO;
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.state, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(ahc_init_rows, dim3((Np + 255) / 256), dim3(256), 0, ctx->stream, w);
    hipLaunchKernelGGL(ahc_transpose, dim3((Np + 31) / 32, (d + 31) / 32), dim3(256), 0, ctx->stream, d_data, w.XT, w.N, w.Np, w.d);
    FA_TRY(exact_rebuild(ctx, w));
    AhcState h{};
    FA_HIP_TRY(ctx, hipMemcpyAsync(&h, w.state, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (h.error) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance");
    if (init.mode == FA_AHC_MODE_AUTO) {
        double dmax;
        const long long bits = static_cast<long long>(h.dmax_bits);
        memcpy(&dmax, &bits, sizeof(dmax));
        // rounding bound of the Lance-Williams recurrence: <= 8 u dmax per merge level (3 products, 2 sums, 3 rounded
        // weights), errors of the two parents enter with weights wa + wb = 1, tree depth <= N; factor 2 of margin.
        const double eps = 16.0 * static_cast<double>(N) * 1.1102230246251565e-16 * dmax;
        FA_HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<char *>(w.state) + offsetof(AhcState, eps), &eps, sizeof(eps), hipMemcpyHostToDevice, ctx->stream));
    }
    FA_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));

    // one graph = kRoundsPerGraph (select, apply) pairs; replayed until the device reports done
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool use_graph = true;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int i = 0; i < kRoundsPerGraph; ++i) {
            hipLaunchKernelGGL(ahc_select, dim3(1), dim3(kSelThreads), 0, ctx->stream, w);
            hipLaunchKernelGGL(ahc_apply, dim3(w.nblk), dim3(kBlk), 0, ctx->stream, w);
        }
        if (hipStreamEndCapture(ctx->stream, &graph) != hipSuccess || !graph) use_graph = false;
        else if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) use_graph = false;
    } else use_graph = false;
    (void)hipGetLastError();
    struct GraphGuard { hipGraph_t &g; hipGraphExec_t &e; ~GraphGuard() { if (e) (void)hipGraphExecDestroy(e); if (g) (void)hipGraphDestroy(g); } } gg{graph, exec};

    long long fallback = 0;
    const long long max_batches = 64 + 8 * static_cast<long long>(N) / kRoundsPerGraph;  // bound on rounds (merges + rescans)
    fa_status st = FA_SUCCESS;
    for (long long it = 0; it < max_batches; ++it) {
        if (use_graph) FA_HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
        else
            for (int i = 0; i < kRoundsPerGraph; ++i) {
                hipLaunchKernelGGL(ahc_select, dim3(1), dim3(kSelThreads), 0, ctx->stream, w);
                hipLaunchKernelGGL(ahc_apply, dim3(w.nblk), dim3(kBlk), 0, ctx->stream, w);
            }
        FA_HIP_TRY(ctx, hipMemcpyAsync(&h, w.state, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h.error == 1) { st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance"); break; }
        if (h.error) { st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: internal selection failure (%d)", h.error); break; }
        if (h.done) break;
        if (h.halt && h.need_exact) {  // ambiguity under the Lance-Williams filter: exact rows from here on
            ++fallback;
            AhcState patch = h;
            patch.halt = 0; patch.need_exact = 0; patch.mode = FA_AHC_MODE_EXACT; patch.eps = 0.0; patch.op = OP_NOOP;
            FA_HIP_TRY(ctx, hipMemcpyAsync(w.state, &patch, sizeof(patch), hipMemcpyHostToDevice, ctx->stream));
            FA_TRY(exact_rebuild(ctx, w));
        }
    }
    if (st == FA_SUCCESS && !h.done) st = fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: round budget exhausted at step %d", h.step);
    if (st != FA_SUCCESS) return st;
    hipLaunchKernelGGL(ahc_finish, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, w);
    FA_HIP_TRY(ctx, hipMemcpyAsync(d_Z, w.Z, sizeof(double) * 4 * (N - 1), hipMemcpyDeviceToDevice, ctx->stream));
    FA_HIP_TRY(ctx, hipEventRecord(ev[2], ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        stats->merges = h.step; stats->rounds = h.rounds; stats->rescans = h.rescans; stats->exact_fallback = fallback;
        stats->windows = h.windows;
        stats->init_ms = t01; stats->merge_ms = t12; stats->total_ms = t01 + t12;
    }
    return FA_SUCCESS;
}

fa_status linkage_checks(const double *data, size_t n, size_t d, double *z, size_t zlen, bool *trivial) {
    // status contract of FastClusterWrapper.cpp:203-226
    *trivial = true;
    if (!data || !z) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (d == 0) return FA_INVALID_ARGUMENT;
    if (n > static_cast<size_t>(INT32_MAX) || d > static_cast<size_t>(INT32_MAX)) return FA_INDEX_OVERFLOW;
    const size_t need = n > 1 ? (n - 1) * 4 : 0;
    if (zlen < need) return FA_OUTPUT_TOO_SMALL;
    if (n == 1) return FA_SUCCESS;
    *trivial = false;
    return FA_SUCCESS;
}

std::mutex g_default_mutex;
fa_ctx *g_default_ctx = nullptr;

}  // namespace

extern "C" {

fa_status fa_ahc_linkage(fa_ctx *ctx, const double *data, size_t n, size_t d, double *dendrogram, size_t dendrogram_len,
                         int32_t mode, int32_t device_pointers, fa_ahc_stats *stats) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    bool trivial;
    const fa_status pre = linkage_checks(data, n, d, dendrogram, dendrogram_len, &trivial);
    if (pre != FA_SUCCESS || trivial) return pre;
    if (stats) memset(stats, 0, sizeof(*stats));
    try {
        fa::DeviceGuard guard(ctx->device);
        if (device_pointers) return ahc_run_device(ctx, data, n, d, dendrogram, mode, stats);
        fa::DevBuf d_in, d_z;
        if (d_in.alloc(sizeof(double) * n * d) != hipSuccess || d_z.alloc(sizeof(double) * 4 * (n - 1)) != hipSuccess) {
            (void)hipGetLastError();
            return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: input staging allocation failed");
        }
        FA_HIP_TRY(ctx, hipMemcpyAsync(d_in.p, data, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream));
        FA_TRY(ahc_run_device(ctx, d_in.as<double>(), n, d, d_z.as<double>(), mode, stats));
        FA_HIP_TRY(ctx, hipMemcpyAsync(dendrogram, d_z.p, sizeof(double) * 4 * (n - 1), hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FA_RUNTIME_ERROR;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fastcluster_wrapper_status fastcluster_compute_centroid_linkage(const double *data, size_t pointCount, size_t dimension,
                                                                double *dendrogramOut, size_t dendrogramLength) {
    bool trivial;
    const fa_status pre = linkage_checks(data, pointCount, dimension, dendrogramOut, dendrogramLength, &trivial);
    if (pre != FA_SUCCESS || trivial) return static_cast<fastcluster_wrapper_status>(pre);
    try {
        std::lock_guard<std::mutex> lock(g_default_mutex);  // re-entrant from any thread; calls are serialised on the one device
        if (!g_default_ctx) {
            int dev = 0;
            if (const char *env = getenv("FLUIDAUDIO_HIP_DEVICE")) dev = atoi(env);
            const fa_status st = fa_ctx_create(dev, nullptr, &g_default_ctx);
            if (st != FA_SUCCESS) return static_cast<fastcluster_wrapper_status>(st == FA_ALLOCATION_FAILURE ? st : FA_RUNTIME_ERROR);
        }
        return static_cast<fastcluster_wrapper_status>(
            fa_ahc_linkage(g_default_ctx, data, pointCount, dimension, dendrogramOut, dendrogramLength, FA_AHC_MODE_AUTO, 0, nullptr));
    } catch (const std::bad_alloc &) {
        return FASTCLUSTER_WRAPPER_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FASTCLUSTER_WRAPPER_RUNTIME_ERROR;
    } catch (...) {
        return FASTCLUSTER_WRAPPER_UNKNOWN_ERROR;
    }
}

fa_status fa_ahc_cut(const double *z, size_t n, double threshold, int32_t *labels) {
    // AHCClustering.swift:112-121 (clamp), :124-197 (top-down cut), :200-210 (relabel by first appearance)
    if (n == 0) return FA_SUCCESS;
    if (!labels || (n > 1 && !z)) return FA_INVALID_ARGUMENT;
    if (n == 1) { labels[0] = 0; return FA_SUCCESS; }
    try {
        double thr = threshold;
        if (thr != thr) thr = 0.0;
        thr = std::max(0.0, std::min(2.0, thr));
        const size_t total = 2 * n - 1;
        std::vector<int64_t> left(total, -1), right(total, -1), assign(n, -1);
        std::vector<double> height(total, 0.0);
        for (size_t r = 0; r + 1 < n; ++r) {
            left[n + r] = static_cast<int64_t>(z[4 * r]);
            right[n + r] = static_cast<int64_t>(z[4 * r + 1]);
            height[n + r] = z[4 * r + 2];
        }
        std::vector<int64_t> stack{static_cast<int64_t>(total - 1)}, queue;
        int64_t next = 0;
        while (!stack.empty()) {
            const int64_t node = stack.back();
            stack.pop_back();
            if (node < 0) continue;
            if (node < static_cast<int64_t>(n)) { if (assign[node] == -1) assign[node] = next++; continue; }
            if (height[node] <= thr) {
                const int64_t label = next++;
                queue.assign(1, node);
                while (!queue.empty()) {
                    const int64_t cur = queue.back();
                    queue.pop_back();
                    if (cur < static_cast<int64_t>(n)) assign[cur] = label;
                    else { if (left[cur] >= 0) queue.push_back(left[cur]); if (right[cur] >= 0) queue.push_back(right[cur]); }
                }
            } else {
                if (left[node] >= 0) stack.push_back(left[node]);
                if (right[node] >= 0) stack.push_back(right[node]);  // popped first => right subtree visited first
            }
        }
        for (size_t i = 0; i < n; ++i) if (assign[i] == -1) assign[i] = next++;
        std::vector<int32_t> remap(static_cast<size_t>(next), -1);
        int32_t nid = 0;
        for (size_t i = 0; i < n; ++i) {
            if (remap[assign[i]] < 0) remap[assign[i]] = nid++;
            labels[i] = remap[assign[i]];
        }
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fa_status fa_ahc_cluster(fa_ctx *ctx, const double *x, size_t n, size_t d, double threshold, int32_t mode, int32_t *labels,
                         fa_ahc_stats *stats) {
    // AHCClustering.swift:20-67
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (!labels) return FA_INVALID_ARGUMENT;
    if (d == 0) { for (size_t i = 0; i < n; ++i) labels[i] = 0; return FA_SUCCESS; }
    if (!x) return FA_INVALID_ARGUMENT;
    if (n == 1) { labels[0] = 0; return FA_SUCCESS; }
    try {
        std::vector<double> norm(n * d), z((n - 1) * 4, 0.0);
        for (size_t i = 0; i < n; ++i) {  // normalizeFeatures (:70-105)
#pragma clang fp contract(off)
            const double *row = x + i * d;
            double ss = 0.0;
            for (size_t k = 0; k < d; ++k) ss += row[k] * row[k];
            const double scale = ss > 0 ? 1.0 / std::sqrt(ss) : 0.0;
            for (size_t k = 0; k < d; ++k) norm[i * d + k] = row[k] * scale;
        }
        const fa_status st = fa_ahc_linkage(ctx, norm.data(), n, d, z.data(), z.size(), mode, 0, stats);
        if (st != FA_SUCCESS) {
            for (size_t i = 0; i < n; ++i) labels[i] = static_cast<int32_t>(i);  // degrade, don't crash (:52-55)
            return st;
        }
        return fa_ahc_cut(z.data(), n, threshold, labels);
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

}  // extern "C"
