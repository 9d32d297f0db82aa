// post.hip — what OfflineDiarizerManager.cluster does with the VBx posteriors (gfx950, fp64).
//
// Replaces computeCentroids (reference: Sources/FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:613-691:
// gamma-weighted mean of the 256-d embeddings for every speaker with pi > 1e-7) and assignEmbeddings (:789-822: cosine
// argmax of every embedding against the centroids, first maximum wins; normalisation :824-859).
// Both are tiny next to AHC/VBx (N ~ 4e4, K <= ~100, d = 256), so the kernels keep the reference's summation ORDER
// (sequential in t for the centroid sums, sequential in the dimension for norms and dot products) and are therefore
// bit-identical to the CPU restatement, not just close: one thread owns one output element and walks the reduction
// axis; the loads of different iterations are independent, so they pipeline.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "fa_common.h"

namespace {

constexpr int kThreads = 256;

// centroid[K][d]: thread (speaker slot c, dimension k).  num[k] += w * emb[t][k] with one rounding per operation (cblas_daxpy :655-672)
__global__ void centroid_kernel(const double *__restrict__ emb, const double *__restrict__ gamma, const int32_t *__restrict__ spk,
                                double *__restrict__ cent, int64_t n, int d, int S, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= d || c >= K) return;
    const int s = spk[c];
    // The sums run in row order (one rounding per operation, like the reference's daxpy), so the ADDITIONS are a dependent chain — but
    // the loads are not: 16 rows are requested at a time (the one-row loop paid a full memory round trip per row: 21 ms at 43 200 rows).
    double num = 0.0, den = 0.0;
    constexpr int kAhead = 16;
    for (int64_t t0 = 0; t0 < n; t0 += kAhead) {
        double wv[kAhead], ev[kAhead];
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            const int64_t t = t0 + j;
            const bool in = t < n;
            wv[j] = in ? gamma[t * S + s] : 0.0;
            ev[j] = in ? emb[t * d + k] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            if (!(wv[j] > 0)) continue;
            den = __dadd_rn(den, wv[j]);
            num = __dadd_rn(num, __dmul_rn(wv[j], ev[j]));
        }
    }
    cent[static_cast<int64_t>(c) * d + k] = den > 0 ? __ddiv_rn(num, den) : 0.0;
}

// The same sums for long recordings: the additions of one (speaker, dimension) are a dependent chain in row order — that is the
// reference's daxpy order and what makes the result bit-identical to the CPU restatement — but nothing says the LOADS have to be:
// centroid_kernel keeps 16 rows in flight per thread and runs on K x d / 64 wavefronts (12 x 4 at the 8 h session: 4.1 ms, every
// 16 rows cost a full memory round trip).  Here a workgroup of 4 wavefronts owns (speaker, 64 dimensions): all four fetch tiles of
// 128 rows (32 requests in flight per thread, coalesced 512-byte rows) into a double-buffered LDS tile while wavefront 0 walks
// the previous tile in row order.  Same operations in the same order: identical bits.
constexpr int kCenTile = 128;
// Round 6 (second half): the kernel was 1.44 ms at the 8 h session and, with one workgroup per CU (133 KB of LDS), 11 of the 17 ms that eight recordings spend
// behind their merge chains in one fa_offline_cluster_batch call.  The chain of one (speaker, dimension) was 9 vector instructions per row: two additions, a
// multiplication, a compare and four selects for the reference's `weight <= 0 -> skip` (:655).  Now:
//   * a row of weight +0 needs no select when its embedding is finite: num + 0 * e == num and den + 0 == den bit for bit (the sums start at +0; -0 + +0 = +0).
//     `rows_finite` (the caller knows: the training rows of the clustering stage are the finite ones) turns the selects off; the fetching wavefronts vote on
//     their weights, and a tile that holds a negative or NaN weight walks the select form — the same bits either way on valid input;
//   * the denominator is the same sum for all 64 dimensions: wavefront 1 walks it while wavefront 0 walks the numerators (one multiplication + one addition per
//     row on the chain).
__global__ __launch_bounds__(256) void centroid_tiled_kernel(const double *__restrict__ emb, const double *__restrict__ gamma, const int32_t *__restrict__ spk,
                                                             double *__restrict__ cent, int64_t n, int d, int S, int K, const int rows_finite) {
    extern __shared__ double cen_lds[];                      // [2][kCenTile][64] values, then [2][kCenTile] weights, then the denominator and [2] tile flags
    double *wbuf = cen_lds + 2 * kCenTile * 64;
    double *s_den = wbuf + 2 * kCenTile;
    int *s_flag = reinterpret_cast<int *>(s_den + 1);        // [2]: the tile in this buffer holds a weight that is negative or NaN
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane, c = blockIdx.y;
    const int s = spk[c];
    const bool kin = k < d;
    constexpr int kPer = kCenTile / 4;                       // rows per wavefront and tile
    double ev[kPer], wv = 0.0;
    auto fetch = [&](const int64_t t0) {                     // rows t0 + wave * kPer + j
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int64_t t = t0 + wave * kPer + j;
            ev[j] = t < n && kin ? emb[t * d + k] : 0.0;
        }
        const int64_t tw = t0 + wave * kPer + lane;          // lane j < kPer carries the weight of row j of this wavefront's share
        wv = lane < kPer && tw < n ? gamma[tw * S + s] : 0.0;
    };
    auto put = [&](const int buf) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) cen_lds[(static_cast<size_t>(buf) * kCenTile + wave * kPer + j) * 64 + lane] = ev[j];
        if (lane < kPer) wbuf[buf * kCenTile + wave * kPer + lane] = wv;
        if (__builtin_amdgcn_ballot_w64(!(wv >= 0.0)) != 0 && lane == 0) atomicOr(&s_flag[buf], 1);
    };
    if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
    __syncthreads();
    double num = 0.0, den = 0.0;
    fetch(0);
    put(0);
    __syncthreads();
    int buf = 0;
    for (int64_t t0 = 0; t0 < n; t0 += kCenTile, buf ^= 1) {
        if (t0 + kCenTile < n) fetch(t0 + kCenTile);         // in flight while wavefronts 0 / 1 add
        const bool plain = rows_finite != 0 && s_flag[buf] == 0;
        if (wave == 0) {
            const double *tile = cen_lds + static_cast<size_t>(buf) * kCenTile * 64 + lane;
            const double *wt = wbuf + buf * kCenTile;
            for (int r0 = 0; r0 < kCenTile; r0 += 16) {   // 32 LDS reads requested together, then 16 branch-free steps of the chain
                double ww[16], ee[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { ww[j] = wt[r0 + j]; ee[j] = tile[(r0 + j) * 64]; }
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) num = __dadd_rn(num, __dmul_rn(ww[j], ee[j]));
                } else {                                   // a skipped row (weight <= 0, :655) keeps the old sum through a select — no 0 * e is ever added
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const double n2 = __dadd_rn(num, __dmul_rn(ww[j], ee[j])); num = ww[j] > 0 ? n2 : num; }
                }
            }
        } else if (wave == 1) {
            const double *wt = wbuf + buf * kCenTile;
            for (int r0 = 0; r0 < kCenTile; r0 += 16) {
                double ww[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) ww[j] = wt[r0 + j];
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) den = __dadd_rn(den, ww[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const double d2 = __dadd_rn(den, ww[j]); den = ww[j] > 0 ? d2 : den; }
                }
            }
        }
        __syncthreads();                                     // the tile and its flag have been read
        if (threadIdx.x == 0) s_flag[buf] = 0;               // (the buffer is filled again two tiles on: behind the next barrier)
        if (t0 + kCenTile < n) put(buf ^ 1);                 // the other buffer: its last readers finished before the barrier above
        __syncthreads();
    }
    if (threadIdx.x == 64) s_den[0] = den;
    __syncthreads();
    den = s_den[0];
    if (wave == 0 && kin) cent[static_cast<int64_t>(c) * d + k] = den > 0 ? __ddiv_rn(num, den) : 0.0;
}

// unit-normalised copy (normalize :824-859: a vector with sum of squares <= 0 is returned unchanged)
__global__ void normalize_rows(const double *__restrict__ v, double *__restrict__ out, int64_t rows, int d) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double *x = v + r * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss = __dadd_rn(ss, __dmul_rn(x[k], x[k]));
    const double scale = ss <= 0 ? 1.0 : __ddiv_rn(1.0, __dsqrt_rn(ss));
    for (int k = 0; k < d; ++k) out[r * d + k] = ss <= 0 ? x[k] : __dmul_rn(x[k], scale);
}

// argmax_k <e_i, c_k> over normalised vectors, first maximum (strict '>', :806-816)
__global__ void assign_kernel(const double *__restrict__ emb, const double *__restrict__ cn, int32_t *__restrict__ out, int64_t n,
                              int d, int K) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *x = emb + i * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss = __dadd_rn(ss, __dmul_rn(x[k], x[k]));
    const double scale = ss <= 0 ? 1.0 : __ddiv_rn(1.0, __dsqrt_rn(ss));
    const bool keep = ss <= 0;
    double best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < K; ++c) {
        const double *cv = cn + static_cast<int64_t>(c) * d;
        double dot = 0.0;
        for (int k = 0; k < d; ++k) {
            const double e = keep ? x[k] : __dmul_rn(x[k], scale);
            dot = __dadd_rn(dot, __dmul_rn(e, cv[k]));
        }
        if (dot > best) { best = dot; bi = c; }
    }
    out[i] = bi;
}

// centroidScores (:789-798): scores[i][k] = <normalised e_i, normalised c_k>, same arithmetic as assign_kernel
__global__ void scores_kernel(const double *__restrict__ emb, const double *__restrict__ cn, double *__restrict__ scores, int64_t n, int d,
                              int K) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *x = emb + i * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss = __dadd_rn(ss, __dmul_rn(x[k], x[k]));
    const double scale = ss <= 0 ? 1.0 : __ddiv_rn(1.0, __dsqrt_rn(ss));
    const bool keep = ss <= 0;
    for (int c = 0; c < K; ++c) {
        const double *cv = cn + static_cast<int64_t>(c) * d;
        double dot = 0.0;
        for (int k = 0; k < d; ++k) {
            const double e = keep ? x[k] : __dmul_rn(x[k], scale);
            dot = __dadd_rn(dot, __dmul_rn(e, cv[k]));
        }
        scores[i * K + c] = dot;
    }
}

// ConstrainedClusterAssignment.assign (reference: Sources/FluidAudio/Diarizer/Offline/Clustering/ConstrainedClusterAssignment.swift:20-42)
// = per segmentation chunk, HungarianAssignment.maxScoreAssignment (Sources/FluidAudio/Diarizer/HungarianAssignment.swift:67-97)
// over the chunk's rows, solved by Kuhn-Munkres with potentials on integer costs (:8-62).  One wavefront per chunk; the
// potentials / matching / slack arrays (n + 1 <= 257 entries) live in the wavefront's LDS slice and the inner loops over
// columns j run across the lanes.  Every comparison is on the reference's integers, so assignments are identical.
constexpr int kHungMaxN = 256;
constexpr long long kHungInf = 0x7fffffffffffffffLL / 4;  // Int.max / 4 (:14)

struct HungLds {
    long long u[kHungMaxN + 1], v[kHungMaxN + 1], minv[kHungMaxN + 1];
    int p[kHungMaxN + 1], way[kHungMaxN + 1];
    unsigned char used[kHungMaxN + 1];
};
// the same arrays as pointers: into the wavefront's LDS slice (n <= kHungMaxN), or into a per-wavefront slab of HBM scratch for
// problems beyond it (more than 256 clusters or rows in a chunk — the reference solves any size, HungarianAssignment.swift:8-62)
struct HungView {
    long long *u, *v, *minv;
    int *p, *way;
    unsigned char *used;
};
__host__ __device__ inline size_t hung_slab_bytes(const int n) {   // per wavefront, 16-byte aligned
    const size_t e = static_cast<size_t>(n) + 1;
    return (e * (3 * sizeof(long long) + 2 * sizeof(int) + 1) + 15) & ~static_cast<size_t>(15);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool BIG>
__global__ __launch_bounds__(256) void hungarian_kernel(const double *__restrict__ scores, const int32_t *__restrict__ chunk_start,
                                                          const int32_t *__restrict__ row_ids, int32_t *__restrict__ out, int n_chunks, int K,
                                                          unsigned char *__restrict__ slabs, int slab_n) {
    __shared__ HungLds lds[BIG ? 1 : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x * 4 + wave;
    if (chunk >= n_chunks) return;
    HungView h;
    if (BIG) {
        unsigned char *base = slabs + static_cast<size_t>(chunk) * hung_slab_bytes(slab_n);
        const size_t e = static_cast<size_t>(slab_n) + 1;
        h.u = reinterpret_cast<long long *>(base); h.v = h.u + e; h.minv = h.v + e;
        h.p = reinterpret_cast<int *>(h.minv + e); h.way = h.p + e;
        h.used = reinterpret_cast<unsigned char *>(h.way + e);
    } else {
        HungLds &l = lds[wave];
        h.u = l.u; h.v = l.v; h.minv = l.minv; h.p = l.p; h.way = l.way; h.used = l.used;
    }
    const int r0 = chunk_start[chunk], R = chunk_start[chunk + 1] - r0;
    const int32_t *rows = row_ids + r0;
    if (K <= 0) { for (int r = lane; r < R; r += 64) out[rows[r]] = -2; return; }  // no columns: every row unassigned (:71)
    const int n = R > K ? R : K;
    // finite range of the chunk's scores (:73-76)
    double mx = -INFINITY, mn = INFINITY;
    for (int e = lane; e < R * K; e += 64) {
        const double sc = scores[static_cast<int64_t>(rows[e / K]) * K + e % K];
        if (isfinite(sc)) { mx = sc > mx ? sc : mx; mn = sc < mn ? sc : mn; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double omx = __shfl_xor(mx, off), omn = __shfl_xor(mn, off);
        mx = omx > mx ? omx : mx; mn = omn < mn ? omn : mn;
    }
    const double max_score = mx == -INFINITY ? 0.0 : mx, min_score = mn == INFINITY ? 0.0 : mn;
    const double sentinel = min_score - 1.0;
    auto cost = [&](const int r, const int c) -> long long {  // (:84-90) padding cells cost 0
        if (r >= R || c >= K) return 0;
        double sc = scores[static_cast<int64_t>(rows[r]) * K + c];
        if (!isfinite(sc)) sc = sentinel;
        return static_cast<long long>(round((max_score - sc) * 1e6));
    };
    for (int j = lane; j <= n; j += 64) { h.u[j] = 0; h.v[j] = 0; h.p[j] = 0; h.way[j] = 0; }
    wave_sync();
    for (int i = 1; i <= n; ++i) {
        for (int j = lane; j <= n; j += 64) { h.minv[j] = kHungInf; h.used[j] = 0; }
        if (lane == 0) h.p[0] = i;
        wave_sync();
        int j0 = 0;
        do {
            if (lane == 0) h.used[j0] = 1;
            wave_sync();
            const int i0 = h.p[j0];
            const long long ui0 = h.u[i0];
            long long delta = kHungInf;
            int j1 = 0x7fffffff;
            for (int j = 1 + lane; j <= n; j += 64) {
                if (h.used[j]) continue;
                const long long cur = cost(i0 - 1, j - 1) - ui0 - h.v[j];
                if (cur < h.minv[j]) { h.minv[j] = cur; h.way[j] = j0; }
                const long long mv = h.minv[j];
                if (mv < delta) { delta = mv; j1 = j; }  // ascending j per lane: first minimum kept (:36-39)
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const long long od = __shfl_xor(delta, off);
                const int oj = __shfl_xor(j1, off);
                if (od < delta || (od == delta && oj < j1)) { delta = od; j1 = oj; }
            }
            if (j1 == 0x7fffffff) j1 = 0;  // cannot happen for n >= 1 (some column is always unused here)
            wave_sync();
            for (int j = lane; j <= n; j += 64) {
                if (h.used[j]) { h.u[h.p[j]] += delta; h.v[j] -= delta; }  // p is injective on the used columns
                else h.minv[j] -= delta;
            }
            wave_sync();
            j0 = j1;
        } while (h.p[j0] != 0);
        if (lane == 0) {
            do { const int j1 = h.way[j0]; h.p[j0] = h.p[j1]; j0 = j1; } while (j0 != 0);
        }
        wave_sync();
    }
    for (int j = 1 + lane; j <= n; j += 64) {
        const int r = h.p[j] - 1;
        if (r >= 0 && r < R) out[rows[r]] = j - 1 < K ? j - 1 : -2;  // dropped slot (:93-96, ConstrainedClusterAssignment.swift:37)
    }
}

}  // namespace

extern "C" {

fa_status fa_vbx_weighted_centroids(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *gamma, const double *pi,
                                    int32_t S, double *centroids, int32_t *map, int32_t *n_centroids) {
    if (!ctx || !n_centroids || !map || (S > 0 && !pi)) return FA_INVALID_ARGUMENT;
    *n_centroids = 0;
    if (n < 0 || d < 1 || S < 0) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "centroids: bad shape");
    try {
        std::vector<int32_t> spk;
        for (int s = 0; s < S; ++s) {  // speakers kept: pi > 1e-7 (:630-640)
            map[s] = -1;
            if (pi[s] > 1e-7) { map[s] = static_cast<int32_t>(spk.size()); spk.push_back(s); }
        }
        const int K = static_cast<int>(spk.size());
        *n_centroids = K;
        if (K == 0) return FA_SUCCESS;
        if (!centroids || (n > 0 && (!emb || !gamma))) return FA_INVALID_ARGUMENT;
        fa::DeviceGuard guard(ctx->device);
        fa::DevBuf d_emb, d_gamma, d_spk, d_cent;
        hipError_t e;
        fa_status st = FA_SUCCESS;
        do {
            if ((e = d_emb.alloc(sizeof(double) * n * d)) != hipSuccess) break;
            if ((e = d_gamma.alloc(sizeof(double) * n * S)) != hipSuccess) break;
            if ((e = d_spk.alloc(sizeof(int32_t) * K)) != hipSuccess) break;
            if ((e = d_cent.alloc(sizeof(double) * K * d)) != hipSuccess) break;
            if (n > 0 && (e = hipMemcpyAsync(d_emb.p, emb, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            if (n > 0 && (e = hipMemcpyAsync(d_gamma.p, gamma, sizeof(double) * n * S, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            if ((e = hipMemcpyAsync(d_spk.p, spk.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            if ((st = fa::centroids_dev(ctx, d_emb.as<double>(), n, d, d_gamma.as<double>(), S, d_spk.as<int32_t>(), K, d_cent.as<double>())) != FA_SUCCESS) break;
            if ((e = hipMemcpyAsync(centroids, d_cent.p, sizeof(double) * K * d, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
            e = hipStreamSynchronize(ctx->stream);
        } while (0);
        if (st != FA_SUCCESS) return st;
        return fa::hip_status(ctx, e, "fa_vbx_weighted_centroids");
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fa_status fa_assign_cosine(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *centroids, int32_t K, int32_t *out) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (n < 0 || d < 1 || K < 0 || !out || !emb) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "assign: bad arguments");
    if (K == 0) { for (int64_t i = 0; i < n; ++i) out[i] = 0; return FA_SUCCESS; }  // guard (:795-797)
    if (!centroids) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_emb, d_c, d_cn, d_out;
    hipError_t e;
    fa_status st = FA_SUCCESS;
    do {
        if ((e = d_emb.alloc(sizeof(double) * n * d)) != hipSuccess) break;
        if ((e = d_c.alloc(sizeof(double) * K * d)) != hipSuccess) break;
        if ((e = d_cn.alloc(sizeof(double) * K * d)) != hipSuccess) break;
        if ((e = d_out.alloc(sizeof(int32_t) * n)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_emb.p, emb, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_c.p, centroids, sizeof(double) * K * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((st = fa::assign_dev(ctx, d_emb.as<double>(), n, d, d_c.as<double>(), K, d_cn.as<double>(), d_out.as<int32_t>())) != FA_SUCCESS) break;
        if ((e = hipMemcpyAsync(out, d_out.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    if (st != FA_SUCCESS) return st;
    return fa::hip_status(ctx, e, "fa_assign_cosine");
}

fa_status fa_centroid_scores(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *centroids, int32_t K, double *scores) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n == 0 || K == 0) return FA_SUCCESS;
    if (n < 0 || d < 1 || K < 0 || !emb || !centroids || !scores) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "scores: bad arguments");
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_emb, d_c, d_cn, d_s;
    hipError_t e;
    fa_status st = FA_SUCCESS;
    do {
        if ((e = d_emb.alloc(sizeof(double) * n * d)) != hipSuccess) break;
        if ((e = d_c.alloc(sizeof(double) * K * d)) != hipSuccess) break;
        if ((e = d_cn.alloc(sizeof(double) * K * d)) != hipSuccess) break;
        if ((e = d_s.alloc(sizeof(double) * n * K)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_emb.p, emb, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_c.p, centroids, sizeof(double) * K * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((st = fa::scores_dev(ctx, d_emb.as<double>(), n, d, d_c.as<double>(), K, d_cn.as<double>(), d_s.as<double>())) != FA_SUCCESS) break;
        if ((e = hipMemcpyAsync(scores, d_s.p, sizeof(double) * n * K, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    if (st != FA_SUCCESS) return st;
    return fa::hip_status(ctx, e, "fa_centroid_scores");
}

fa_status fa_constrained_assign(fa_ctx *ctx, const double *scores, int64_t n, int32_t K, const int32_t *chunk_indices, int32_t *out) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (n < 0 || n > INT32_MAX || K < 0 || !chunk_indices || !out || (K > 0 && !scores)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "constrained assign: bad arguments");
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_s, d_out;
    hipError_t e;
    fa_status st = FA_SUCCESS;
    do {
        if ((e = d_s.alloc(sizeof(double) * n * (K > 0 ? K : 1))) != hipSuccess) break;
        if ((e = d_out.alloc(sizeof(int32_t) * n)) != hipSuccess) break;
        if (K > 0 && (e = hipMemcpyAsync(d_s.p, scores, sizeof(double) * n * K, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((st = fa::constrained_assign_dev(ctx, d_s.as<double>(), n, K, chunk_indices, d_out.as<int32_t>())) != FA_SUCCESS) break;
        if ((e = hipMemcpyAsync(out, d_out.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    if (st != FA_SUCCESS) return st;
    return fa::hip_status(ctx, e, "fa_constrained_assign");
}

}  // extern "C"

// ---- device-level cores (fa_common.h): inputs and outputs are device pointers, work is enqueued on ctx->stream
fa_status fa::centroids_dev(fa_ctx *ctx, const double *d_emb, int64_t n, int32_t d, const double *d_gamma, int32_t S, const int32_t *d_spk, int32_t K,
                            double *d_cent, const bool rows_finite) {
    if (K <= 0) return FA_SUCCESS;
    if (n >= 4 * kCenTile && !fa::sw_on(fa::Sw::CENTROID_SIMPLE)) {
        const size_t lds = sizeof(double) * (2 * kCenTile * 64 + 2 * kCenTile + 2);   // + the denominator and the two tile flags
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(centroid_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL(centroid_tiled_kernel, dim3((d + 63) / 64, K), dim3(256), lds, ctx->stream, d_emb, d_gamma, d_spk, d_cent, n, d, S, K, rows_finite ? 1 : 0);
    } else
        hipLaunchKernelGGL(centroid_kernel, dim3((d + 63) / 64, K), dim3(64), 0, ctx->stream, d_emb, d_gamma, d_spk, d_cent, n, d, S, K);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

fa_status fa::scores_dev(fa_ctx *ctx, const double *d_emb, int64_t n, int32_t d, const double *d_cent, int32_t K, double *d_cn, double *d_scores) {
    if (n <= 0 || K <= 0) return FA_SUCCESS;
    hipLaunchKernelGGL(normalize_rows, dim3((K + 63) / 64), dim3(64), 0, ctx->stream, d_cent, d_cn, static_cast<int64_t>(K), d);
    hipLaunchKernelGGL(scores_kernel, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_emb, d_cn, d_scores, n, d, K);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

fa_status fa::assign_dev(fa_ctx *ctx, const double *d_emb, int64_t n, int32_t d, const double *d_cent, int32_t K, double *d_cn, int32_t *d_out) {
    if (n <= 0) return FA_SUCCESS;
    if (K <= 0) { FA_HIP_TRY(ctx, hipMemsetAsync(d_out, 0, sizeof(int32_t) * n, ctx->stream)); return FA_SUCCESS; }   // guard (:795-797)
    hipLaunchKernelGGL(normalize_rows, dim3((K + 63) / 64), dim3(64), 0, ctx->stream, d_cent, d_cn, static_cast<int64_t>(K), d);
    hipLaunchKernelGGL(assign_kernel, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_emb, d_cn, d_out, n, d, K);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

// chunk_indices is a HOST array (it is the caller's bookkeeping, not a device product): the grouping of rows by chunk is host
// work, the tables go up (8 n bytes) and stay alive until the kernel has run (the call synchronises the stream before it returns).
fa_status fa::constrained_assign_dev(fa_ctx *ctx, const double *d_scores, int64_t n, int32_t K, const int32_t *chunk_indices, int32_t *d_out) {
    if (n <= 0) return FA_SUCCESS;
    try {
        // rows grouped by chunk, ascending row order inside a chunk (rowsByChunk[chunk].append(row), :27-30)
        std::vector<int32_t> order(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; ++i) order[i] = static_cast<int32_t>(i);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return chunk_indices[a] < chunk_indices[b]; });
        std::vector<int32_t> starts;
        int max_rows = 0;
        for (int64_t i = 0; i < n; ++i)
            if (i == 0 || chunk_indices[order[i]] != chunk_indices[order[i - 1]]) starts.push_back(static_cast<int32_t>(i));
        starts.push_back(static_cast<int32_t>(n));
        const int n_chunks = static_cast<int>(starts.size()) - 1;
        for (int c = 0; c < n_chunks; ++c) max_rows = std::max(max_rows, starts[c + 1] - starts[c]);
        const int n_max = std::max(max_rows, static_cast<int>(K));
        const bool big = n_max > kHungMaxN;   // potentials / matching in HBM slabs instead of LDS (rare: more than 256 clusters survive VBx)
        fa::DevBuf d_start, d_rows, d_slabs;
        if (big && d_slabs.alloc(ctx, hung_slab_bytes(n_max) * static_cast<size_t>(n_chunks)) != hipSuccess) {
            (void)hipGetLastError();
            return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "constrained assign: device allocation failed");
        }
        if (d_start.alloc(ctx, sizeof(int32_t) * starts.size()) != hipSuccess || d_rows.alloc(ctx, sizeof(int32_t) * n) != hipSuccess) {
            (void)hipGetLastError();
            return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "constrained assign: device allocation failed");
        }
        FA_HIP_TRY(ctx, hipMemcpyAsync(d_start.p, starts.data(), sizeof(int32_t) * starts.size(), hipMemcpyHostToDevice, ctx->stream));
        FA_HIP_TRY(ctx, hipMemcpyAsync(d_rows.p, order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
        FA_HIP_TRY(ctx, hipMemsetAsync(d_out, 0xfe, sizeof(int32_t) * n, ctx->stream));  // placeholder, every row is written
        if (big) hipLaunchKernelGGL(hungarian_kernel<true>, dim3((n_chunks + 3) / 4), dim3(256), 0, ctx->stream, d_scores, d_start.as<int32_t>(), d_rows.as<int32_t>(), d_out, n_chunks, K, d_slabs.as<unsigned char>(), n_max);
        else hipLaunchKernelGGL(hungarian_kernel<false>, dim3((n_chunks + 3) / 4), dim3(256), 0, ctx->stream, d_scores, d_start.as<int32_t>(), d_rows.as<int32_t>(), d_out, n_chunks, K, static_cast<unsigned char *>(nullptr), 0);
        FA_HIP_TRY(ctx, hipGetLastError());
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the tables above are released on return
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}
