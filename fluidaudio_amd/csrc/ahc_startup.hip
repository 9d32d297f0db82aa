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
// How this file computes the same thing.  The N-1 merges are a strictly serial
// chain, so the design minimises the latency of ONE merge: one kernel per round, replayed from a
// hipGraph (a dependent kernel boundary costs ~1.7 us on MI355X, a software grid barrier 4-7 us).
//   * slots: a merged cluster keeps the lower slot; node[slot] = dendrogram node id living there
//     (0..N-1 points, N+s the cluster made by merge s; DEAD once merged away);
//   * centroids are stored append-only by node id, C[node][d] (never overwritten, so every
//     workgroup may read them while one workgroup appends);
//   * M (slot x slot, fp64, N^2*8 B = 20 GB at N = 50 000) is resident in HBM and ASYMMETRIC:
//     the distance of slots (x, y) is valid at M[x][y] iff node[x] > node[y] (the row of the more
//     recently created cluster).  A merge therefore only rewrites ONE row (coalesced); no column is
//     ever written and dead columns need no clean-up;
//   * per row x: d1[x] = minimum over all other active slots, nn[x] its slot (lowest on ties) or -1
//     when the nearest neighbour was merged away — d1 then stays a LOWER BOUND and the row is
//     re-scanned only when that bound reaches the global minimum (the reference's lazy scheme);
//   * per 256-row block a record (three smallest d1, rows + neighbours of the first two), double
//     buffered by round parity.  Every workgroup starts a round by reducing the same records, so
//     all of them reach the same decision without any inter-workgroup synchronisation inside the
//     round; the kernel boundary is the only barrier;
//   * FA_AHC_MODE_AUTO fills the new row with the Lance-Williams centroid update (O(N) per merge);
//     the pair to merge is taken from those values only when it is the unique mutual-nearest pair
//     with every other row minimum farther than 2*eps (eps = rounding bound of the recurrence);
//     otherwise all matrix entries inside the window are re-evaluated with the reference's exact
//     sum (COLLECT -> PAIRS -> evaluate rounds); an exact tie there, or a window that overflows
//     (massive ties, duplicated inputs), sends the problem to the reference-order run.  Heights are always
//     recomputed after the loop from the stored centroids with the reference's sequential sum, so
//     the merge order never depends on the approximation and the output rows are bit-identical;
//   * FA_AHC_MODE_EXACT: every new-row entry is the reference's sequential fp64 sum (O(N d) per merge).
// Exactly tied distances: the round kernel's order is (value, row, column); the reference's is decided by its binary heap.  A run in
// FA_AHC_MODE_AUTO that meets an exact tie at the minimum (or a window overflowing with near-ties) is therefore recomputed in the
// reference's selection order (ahc_reforder.h, the ro_* kernels below): the output equals the reference row for row on tied input too.
// FA_AHC_MODE_EXACT keeps (value, row, column): same heights and partitions on duplicates, possibly other rows.
// (this unit: the start-up kernels)
#include "ahc_ws.h"

using namespace fa_ahc;

namespace {
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
    if (i < 2 * w.N) w.sizes[i] = 1.0;
    if (i >= w.Np) return;
    w.node[i] = i < w.N ? i : kDead;
    RowSt r; r.d1 = dinf(); r.nn = -1; r.nnnode = -1;
    w.row[i] = r;
    w.e2[i] = dinf();
}

// Initial state, window counters and flags written ON the device, and eps from the maxima the start-up kernels found: the set-up of a
// problem needs no host round trip (round 2 read dmax / nmax back, computed eps on the host and uploaded it: two stream synchronisations
// per call — most of the fixed cost of a short recording).  A NaN met by the start-up kernels sets flags[0]; the first round halts on it.
__global__ void ahc_init_state(Ws w, int mode) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    AhcState s{};
    s.mode = mode;
    for (int k = 0; k < kPend; ++k) { s.pend_row[k] = -1; s.pend_node[k] = -1; }
    s.prev_op = OP_NONE;
    s.n_points = w.N; s.rounds32 = 0;
    s.sym_limit = w.N;                                     // the start-up writes the full matrix: every pair of points has both copies
    w.state[0] = s; w.state[1] = s;
    for (int i = 0; i < 4; ++i) { w.cnt[i].stale_key = ~0ULL; w.cnt[i].ncand = 0; w.cnt[i].npairs = 0; }
    for (int i = 0; i < 4; ++i) w.flags[i] = 0;
    for (int i = 0; i < 16; ++i) w.prof[i] = 0;
}

__global__ void ahc_set_eps(Ws w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const AhcState s = w.state[0];
    double eps = 0.0;
    if (s.mode == FA_AHC_MODE_AUTO) {
        // rounding bound of the Lance-Williams recurrence: <= 9.5 u dmax per merge level — weights from one reciprocal (wa, wb: 2 u each,
        // wab: 5 u), 3 products, 2 sums: (3 u)(wa da + wb db) + (6 u) wab dab + 2 u dmax <= (3 + 1.5 + 2) u dmax, plus the tree-summed
        // d(a,b) (~10 ulp of it, weighted by wab <= 1/4: 2.5 u dmax); errors of the two parents enter with weights wa + wb = 1, tree
        // depth <= N; 16 u per level leaves a margin of 1.7.
        // Start-up matrix in Gram form: |x|^2 + |y|^2 - 2 x.y carries <= (d + 2) u (|x|^2 + |y|^2 + 2 |x||y|) <= 4 (d + 2) u nmax.
        const double dmax = __longlong_as_double(static_cast<long long>(s.dmax_bits)), nmax = __longlong_as_double(static_cast<long long>(s.nmax_bits));
        const double u = 1.1102230246251565e-16;
        eps = 16.0 * static_cast<double>(w.N) * u * dmax + 8.0 * (static_cast<double>(w.d) + 2.0) * u * nmax;
    }
    w.state[0].eps = eps; w.state[1].eps = eps;
}

// Exact pairwise squared distances of the live slots, the reference's summation order
// (FastClusterWrapper.cpp:45-52).  Both triangles are written; dead slots and the diagonal get +inf.
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
        const bool ai = w.node[i] != kDead;
        double out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx * 4 + c;
            const bool ok = ai && i != j && w.node[j] != kDead;
            const double v = acc[r][c];
            if (ok) { if (v != v) bad = true; else if (v > lmax) lmax = v; }
            out[c] = ok ? v : dinf();
        }
        double *dst = w.M + static_cast<size_t>(i) * Np + j0 + tx * 4;
        reinterpret_cast<double2 *>(dst)[0] = make_double2(out[0], out[1]);
        reinterpret_cast<double2 *>(dst)[1] = make_double2(out[2], out[3]);
    }
    if (bad) w.flags[0] = 1;  // nan_error (FastClusterWrapper.cpp:60-62)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(lmax, off); if (o > lmax) lmax = o; }
    if ((tid & 63) == 0 && lmax > 0.0)
        atomicMax(&w.state[0].dmax_bits, static_cast<unsigned long long>(__double_as_longlong(lmax)));
}

// ---- FA_AHC_MODE_AUTO start: the N x d . d x N contraction on the fp64 matrix cores ---------------------------------
// In AUTO mode every matrix entry is only a filter (decisions inside 2 eps are re-evaluated exactly), so the initial
// matrix may be the Gram form |x|^2 + |y|^2 - 2 x.y: 2 N^2 d = 1.28 TFLOP at N = 50 000, d = 256 on
// v_mfma_f64_16x16x4_f64 instead of 1.9 T dependent fp64 VALU operations.  Its rounding error (<= ~(d + 2) u (|x| + |y|)^2)
// is added to eps by the host.  Workgroup = 128 x 128 tile, wavefront = 64 x 64 (4 x 4 MFMA tiles, 64 accumulator
// doubles per lane); operands staged k-major in LDS with a 144-double row stride (two k rows of a 32-lane ds_read_b64
// service group land 32 banks apart).
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int GK = 16, GS = 144;   // (GT, the tile edge: ahc_ws.h)

// Squared norms of the slots (Gram form only: the entries are a filter, their rounding error is inside eps).  Workgroup = 64 slots x 4 quarters of the
// coordinates; the quarters are added in a fixed order (deterministic).  One thread per slot walking all d coordinates was 73 us of dependent loads at
// 43 200 x 256 — 0.7 % of the start-up for 44 MB of reads.
__global__ __launch_bounds__(256) void ahc_sqnorms(Ws w, double *__restrict__ norms) {
    __shared__ double s_q[3][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, x = blockIdx.x * 64 + lane;
    const int per = (w.d + 3) / 4, k0 = q * per, k1 = k0 + per < w.d ? k0 + per : w.d;
    double s = 0.0;
    if (x < w.Np)
        for (int k = k0; k < k1; ++k) { const double v = w.XT[static_cast<size_t>(k) * w.Np + x]; s += v * v; }
    if (q) s_q[q - 1][lane] = s;
    __syncthreads();
    if (q || x >= w.Np) return;
    s = ((s + s_q[0][lane]) + s_q[1][lane]) + s_q[2][lane];
    norms[x] = w.node[x] != kDead ? s : -1.0;   // an empty slot carries a negative "norm": the Gram tiles test liveness on the value they load anyway (no second load per row)
    if (s > 0.0) atomicMax(&w.state[0].nmax_bits, static_cast<unsigned long long>(__double_as_longlong(s)));
}

__global__ __launch_bounds__(256, 2) void ahc_gram_mfma(Ws w, const double *__restrict__ norms) {
    __shared__ double sA[GK][GS];
    __shared__ double sB[GK][GS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the matrix is symmetric: only the tiles on and below the diagonal are computed, an off-diagonal tile is written twice (as it is
    // and mirrored) — half of the 1.28 TFLOP
    if (blockIdx.x > blockIdx.y) return;
    const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
    const bool mirror = i0 != j0;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    const int Np = w.Np, d = w.d;
    v4f64 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = v4f64{0.0, 0.0, 0.0, 0.0};
    // [GK][128] doubles of both operand tiles per k chunk: thread -> (k row, 16-byte column pair); the next chunk travels
    // from L2/HBM into registers while the matrix cores work on the current one
    constexpr int kVec = (GK * GT / 2) / 256;  // 4
    double2 ra[kVec], rb[kVec];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int e = 0; e < kVec; ++e) {
            const int q = tid + 256 * e, k = k0 + q / (GT / 2), c2 = (q % (GT / 2)) * 2;
            ra[e] = rb[e] = make_double2(0.0, 0.0);
            if (k < d) {
                ra[e] = *reinterpret_cast<const double2 *>(w.XT + static_cast<size_t>(k) * Np + i0 + c2);
                rb[e] = *reinterpret_cast<const double2 *>(w.XT + static_cast<size_t>(k) * Np + j0 + c2);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < d; k0 += GK) {
#pragma unroll
        for (int e = 0; e < kVec; ++e) {
            const int q = tid + 256 * e, kk = q / (GT / 2), c2 = (q % (GT / 2)) * 2;
            *reinterpret_cast<double2 *>(&sA[kk][c2]) = ra[e];
            *reinterpret_cast<double2 *>(&sB[kk][c2]) = rb[e];
        }
        __syncthreads();
        if (k0 + GK < d) fetch(k0 + GK);
#pragma unroll 2
        for (int ks = 0; ks < GK / 4; ++ks) {
            const int kr = 4 * ks + (lane >> 4);
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t] = sA[kr][wr + 16 * t + (lane & 15)]; b[t] = sB[kr][wc + 16 * t + (lane & 15)]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[c], acc[r][c], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double lmax = 0.0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_barrier(0);   // one column strip at a time: hoisting the loads of all 64 outputs costs 50 spilled registers at 2 waves per SIMD
        const int j = j0 + wc + 16 * c + (lane & 15);
        const bool lj = w.node[j] != kDead;
        const double nj = norms[j];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + wr + 16 * r + (lane >> 4) + 4 * e;
                const bool ok = lj && i != j && w.node[i] != kDead;
                double v = norms[i] + nj - 2.0 * acc[r][c][e];
                if (ok) { if (v != v) bad = true; }
                if (!(v > 0.0)) v = 0.0;  // duplicates can come out slightly negative; keeps -0.0 out of the bit-pattern reductions
                if (ok && v > lmax) lmax = v;
                w.M[static_cast<size_t>(i) * Np + j] = ok ? v : dinf();
#ifndef FA_GRAM_NO_MIRROR
                if (mirror) w.M[static_cast<size_t>(j) * Np + i] = ok ? v : dinf();   // 4 consecutive doubles per row and store; the four e complete the lines in L2
#endif
            }
    }
    if (bad) w.flags[0] = 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(lmax, off); if (o > lmax) lmax = o; }
    if (lane == 0 && lmax > 0.0)
        atomicMax(&w.state[0].dmax_bits, static_cast<unsigned long long>(__double_as_longlong(lmax)));
}

// The same contraction for d % 16 == 0 (every embedding size on the path), re-plumbed around what scripts/ubench/mfma64.hip measured
// (profiles/r03_ubench_mfma64.txt): a stream of independent v_mfma_f64_16x16x4_f64 runs at 65 TFLOP/s, with eight ds_read_b64 in front of
// every 16 of them at 45, with four ds_read_b128 at 57 — every instruction that WRITES VGPRs while the matrix core runs costs 40-70 of its
// clocks, wherever it is placed and however many wavefronts share the SIMD.  So: operand tiles travel global -> LDS without touching
// registers (global_load_lds_dwordx4: one instruction = one 1 KB k-row of a 128-wide tile; the kernel above needs 8 loads + 8 ds_write per
// thread and chunk for it), two LDS stages and ONE barrier per k-chunk, operands read as 16-byte pairs (rows 2 m, 2 m + 1 of a 32-row
// group -> two MFMA tiles per read: the tile index of the rows is interleaved, which the epilogue undoes), and the results leave as
// 16-byte stores in both the direct and the mirrored direction.
typedef double d2f64 __attribute__((ext_vector_type(2)));
constexpr int G2K = 16, G2S = 144;
constexpr size_t kGram2LdsBytes = sizeof(double) * 2 * 2 * G2K * G2S;   // [stage][operand][k][144]: 73 728 B, two workgroups per CU

// MINIMA (round 6): the tile also leaves, for each of its 128 rows, the minimum / lowest-index argmin / second minimum over its 128 columns —
// and, mirrored, for each of its columns over its rows — in part_vs / part_ix[tile column][row]: ahc_row_minima_parts merges the Np / 128 partials
// of a row.  ahc_row_minima re-read the whole matrix for the same three numbers (15 GB at 43 200 points: 2.8 of the start-up's 13.2 ms).
// Partials are (v, s, i): smallest entry, the smallest entry OTHER than the one at i, lowest index of v; merged by (value, index), the loser's v
// competing for s — associative and commutative, so any tile order gives what one ascending scan gives.
struct MinAcc { double v, s; int i; };
// (entries are non-negative or +inf here — a NaN entry sets the flag that declines the run.  v_min_f64 / v_max_f64 directly: fmin / fmax
// come with a canonicalising v_max_f64 x, x per operand under IEEE mode, a third of the epilogue's instructions when it was written with them)
__device__ __forceinline__ double vmin64(const double a, const double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax64(const double a, const double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void min_ins(MinAcc &a, const double m, const int x) {   // x ascending within one accumulator: the lowest index of equal values is kept
    a.i = m < a.v ? x : a.i;
    a.s = vmin64(a.s, vmax64(a.v, m));
    a.v = vmin64(a.v, m);
}
__device__ __forceinline__ void min_merge(MinAcc &a, const MinAcc o) {
    const bool take = o.v < a.v || (o.v == a.v && o.i < a.i);
    a.s = vmin64(vmin64(a.s, o.s), vmax64(a.v, o.v));   // the loser's minimum competes for the second place
    a.v = vmin64(a.v, o.v);
    a.i = take ? o.i : a.i;
}
__device__ __forceinline__ MinAcc min_xor(const MinAcc a, const int mask) {
    MinAcc o; o.v = __shfl_xor(a.v, mask); o.s = __shfl_xor(a.s, mask); o.i = __shfl_xor(a.i, mask);
    return o;
}

#define FA_GRAM_GRID(T) dim3(static_cast<unsigned>((T) * ((T) + 1) / 2))   // tiles on and below the diagonal
template <bool MINIMA>
__global__ __launch_bounds__(256, 2) void ahc_gram_mfma2_t(Ws w, const double *__restrict__ norms, double2 *__restrict__ part_vs, int *__restrict__ part_ix) {
    extern __shared__ __attribute__((aligned(16))) double sg[];
    // symmetric: one workgroup per tile on or below the diagonal, off-diagonal tiles are written twice.  (Until round 6 a square grid whose upper half returned
    // at once: 57 000 workgroups of 73 KB LDS that each take a slot for a moment — 10.70 -> 10.56 ms at 43 200 points.)  Tile (by, bx), bx <= by, of the linear id:
    int by = static_cast<int>((sqrtf(8.0f * static_cast<float>(blockIdx.x) + 1.0f) - 1.0f) * 0.5f);
    while ((by + 1) * (by + 2) / 2 <= static_cast<int>(blockIdx.x)) ++by;      // (the float root is off by at most one)
    while (by * (by + 1) / 2 > static_cast<int>(blockIdx.x)) --by;
    const int bx = static_cast<int>(blockIdx.x) - by * (by + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;   // wave: in an SGPR, so the row addresses below are scalar
    const int i0 = by * GT, j0 = bx * GT;
    const bool mirror = i0 != j0;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    const int Np = w.Np, nchunk = w.d / G2K;
    v4f64 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = v4f64{0.0, 0.0, 0.0, 0.0};
    // wavefront `wave` moves the k rows 4 wave .. 4 wave + 3 of both operands of a chunk: 8 instructions, 1 KB each
    auto issue = [&](const int k0, const int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = wave * 4 + q;
            const double *row = w.XT + static_cast<size_t>(k0 + kk) * Np;   // wave-uniform
            __builtin_amdgcn_global_load_lds(row + i0 + 2 * lane, sg + ((st * 2 + 0) * G2K + kk) * G2S, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(row + j0 + 2 * lane, sg + ((st * 2 + 1) * G2K + kk) * G2S, 16, 0, 0);
        }
    };
    issue(0, 0);
    for (int c = 0; c < nchunk; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's rows of chunk c are in LDS ...
        __syncthreads();                                    // ... everybody's are, and nobody reads the other stage any more
        if (c + 1 < nchunk) issue((c + 1) * G2K, (c + 1) & 1);
        const double *A = sg + ((c & 1) * 2 + 0) * G2K * G2S + wr + 2 * l15, *B = sg + ((c & 1) * 2 + 1) * G2K * G2S + wc + 2 * l15;
#pragma unroll
        for (int ks = 0; ks < G2K / 4; ++ks) {
            const int kr = 4 * ks + lq;
            d2f64 a2[2], b2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { a2[t] = *reinterpret_cast<const d2f64 *>(A + kr * G2S + 32 * t); b2[t] = *reinterpret_cast<const d2f64 *>(B + kr * G2S + 32 * t); }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    acc[r][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64((r & 1) ? a2[r >> 1].y : a2[r >> 1].x, (cc & 1) ? b2[cc >> 1].y : b2[cc >> 1].x, acc[r][cc], 0, 0, 0);
        }
    }
    // tile (r, cc), register e, lane: row i = i0 + wr + 32 (r >> 1) + 2 (lq + 4 e) + (r & 1), column j = j0 + wc + 32 (cc >> 1) + 2 l15 + (cc & 1)
    double lmax = 0.0;
    bool bad = false;
    auto entry = [&](const double dot, const double ni, const double nj, const bool ok) {
        double v = ni + nj - 2.0 * dot;
        if (ok && v != v) bad = true;
        if (!(v > 0.0)) v = 0.0;  // duplicates can come out slightly negative; keeps -0.0 out of the bit-pattern reductions
        if (ok && v > lmax) lmax = v;
        return ok ? v : dinf();
    };
    // MINIMA: the wave's entries of a strip (64 rows x 32 columns) also go through a private 17 KB piece of the idle operand LDS, row-major with a
    // 34-double stride, and come back transposed: lane l reads ROW l (16 ds_read_b128, bank-conflict free at that stride) and folds its 32 entries
    // in ascending column order into the lane's row accumulator, which simply carries on through the second strip; lanes (c, h) read COLUMN c over
    // the rows 32 h .. 32 h + 31 in ascending order and the two halves meet through one exchange.  No cross-lane reduction per row: a DPP butterfly per row
    // pair and strip cost 5 000 VALU instructions per tile (12.6 instead of 10.3 ms for the kernel: nothing of it hid under the other workgroup's
    // matrix-core loop); this form costs ~1 000.  The norms of the tile's rows and columns are staged in LDS as well.
    MinAcc mine, cmine;
    mine.v = cmine.v = dinf(); mine.s = cmine.s = dinf(); mine.i = cmine.i = INT_MAX;
    constexpr int TS = 34;                                 // doubles per LDS row: 272 B, a multiple of 16 that walks the banks
    double *const tile = sg + wave * 64 * TS;              // [64][TS] of this wave
    double *const sn = sg + 4 * 64 * TS;                   // [0, 128): norms of the rows i0 .., [128, 256): of the columns j0 ..
    static_assert((4 * 64 * TS + 2 * GT) * sizeof(double) <= kGram2LdsBytes, "the epilogue's LDS lives inside the operand stages");
    if constexpr (MINIMA) {
        __syncthreads();     // everybody is done with the operands of the last chunk
        sn[tid] = norms[(tid < GT ? i0 : j0 - GT) + tid];
        __syncthreads();
    }
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {   // column pair group: columns jb, jb + 1
        __builtin_amdgcn_sched_barrier(0);   // one strip at a time (register pressure)
        const int jb = j0 + wc + 32 * cp + 2 * l15;
        double nj0, nj1;
        if constexpr (MINIMA) { const d2f64 q = *reinterpret_cast<const d2f64 *>(sn + GT + wc + 32 * cp + 2 * l15); nj0 = q.x; nj1 = q.y; }
        else { nj0 = norms[jb]; nj1 = norms[jb + 1]; }
        const bool lj0 = !(nj0 < 0.0), lj1 = !(nj1 < 0.0);   // (NaN norms are live rows)
#pragma unroll
        for (int rp = 0; rp < 2; ++rp)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int il = 32 * rp + 2 * (lq + 4 * e), ib = i0 + wr + il;   // rows ib (tiles 2 rp) and ib + 1 (tiles 2 rp + 1); il: within the wave
                double ni0, ni1;
                if constexpr (MINIMA) { const d2f64 q = *reinterpret_cast<const d2f64 *>(sn + wr + il); ni0 = q.x; ni1 = q.y; }
                else { ni0 = norms[ib]; ni1 = norms[ib + 1]; }
                const bool li0 = !(ni0 < 0.0), li1 = !(ni1 < 0.0);
                const double v00 = entry(acc[2 * rp][2 * cp][e], ni0, nj0, li0 && lj0 && ib != jb);
                const double v01 = entry(acc[2 * rp][2 * cp + 1][e], ni0, nj1, li0 && lj1 && ib != jb + 1);
                const double v10 = entry(acc[2 * rp + 1][2 * cp][e], ni1, nj0, li1 && lj0 && ib + 1 != jb);
                const double v11 = entry(acc[2 * rp + 1][2 * cp + 1][e], ni1, nj1, li1 && lj1 && ib + 1 != jb + 1);
                *reinterpret_cast<d2f64 *>(w.M + static_cast<size_t>(ib) * Np + jb) = d2f64{v00, v01};
                *reinterpret_cast<d2f64 *>(w.M + static_cast<size_t>(ib + 1) * Np + jb) = d2f64{v10, v11};
                if (mirror) {
                    *reinterpret_cast<d2f64 *>(w.M + static_cast<size_t>(jb) * Np + ib) = d2f64{v00, v10};
                    *reinterpret_cast<d2f64 *>(w.M + static_cast<size_t>(jb + 1) * Np + ib) = d2f64{v01, v11};
                }
                if constexpr (MINIMA) {
                    *reinterpret_cast<d2f64 *>(tile + il * TS + 2 * l15) = d2f64{v00, v01};
                    *reinterpret_cast<d2f64 *>(tile + (il + 1) * TS + 2 * l15) = d2f64{v10, v11};
                }
            }
        if constexpr (MINIMA) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int jc = j0 + wc + 32 * cp;
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {   // row `lane` of the wave, columns jc .. jc + 31 ascending (a real loop: unrolled whole, its 48 loads are hoisted and spill)
                const d2f64 q = *reinterpret_cast<const d2f64 *>(tile + lane * TS + 2 * k);
                min_ins(mine, q.x, jc + 2 * k);
                min_ins(mine, q.y, jc + 2 * k + 1);
            }
            const int c = lane & 31, h = lane >> 5;
            MinAcc ca;
            ca.v = ca.s = dinf(); ca.i = INT_MAX;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) min_ins(ca, tile[(32 * h + r) * TS + c], i0 + wr + 32 * h + r);   // column c of the strip, rows ascending
            min_merge(ca, min_xor(ca, 32));
            if (h == cp) cmine = ca;         // lane l keeps column l of the wave's 64
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the next strip overwrites the piece
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if constexpr (MINIMA) {
        // the two waves that share rows (wc 0 / 64) and the two that share columns (wr 0 / 64) meet through the sender's own piece
        MinAcc *const xw = reinterpret_cast<MinAcc *>(tile);           // [0, 64): rows, [64, 128): columns
        if (wc) xw[lane] = mine;
        if (wr) xw[64 + lane] = cmine;
        __syncthreads();
        const size_t npz = static_cast<size_t>(Np);
        if (!wc) {
            min_merge(mine, reinterpret_cast<const MinAcc *>(sg + (wave + 1) * 64 * TS)[lane]);          // wave (wr, 64) = this wave + 1
            const size_t at = bx * npz + i0 + wr + lane;
            part_vs[at] = make_double2(mine.v, mine.s);
            part_ix[at] = mine.i;
        }
        if (!wr && mirror) {
            min_merge(cmine, reinterpret_cast<const MinAcc *>(sg + (wave + 2) * 64 * TS)[64 + lane]);    // wave (64, wc) = this wave + 2
            const size_t at = by * npz + j0 + wc + lane;
            part_vs[at] = make_double2(cmine.v, cmine.s);
            part_ix[at] = cmine.i;
        }
    }
    if (bad) w.flags[0] = 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(lmax, off); if (o > lmax) lmax = o; }
    if (lane == 0 && lmax > 0.0)
        atomicMax(&w.state[0].dmax_bits, static_cast<unsigned long long>(__double_as_longlong(lmax)));
}

// Row minimum + lowest-index argmin of every live row of a freshly rebuilt (symmetric) matrix.
__global__ __launch_bounds__(kBlk) void ahc_row_minima(Ws w) {
    __shared__ double s_val[kWaves];
    __shared__ int s_idx[kWaves];
    __shared__ double s_second[kWaves];
    __shared__ int s_win;
    const int i = blockIdx.x;
    double v = dinf(), v2 = dinf();   // the thread's smallest and second smallest entry
    int ix = INT_MAX;
    if (w.node[i] != kDead) {
        const double *row = w.M + static_cast<size_t>(i) * w.Np;
        for (int x = threadIdx.x; x < w.Np; x += kBlk) {
            const double m = row[x];
            if (m < v) { v2 = v; v = m; ix = x; }  // x ascending per thread => lowest index kept
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
    // second smallest entry of the row = the smallest one that is not the winner's (the winner's thread contributes its own second)
    double c2 = mine_ix == s_win ? v2 : mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(c2, off); if (o < c2) c2 = o; }
    if ((threadIdx.x & 63) == 0) s_second[threadIdx.x >> 6] = c2;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kWaves; ++wv) if (s_second[wv] < c2) c2 = s_second[wv];
        w.e2[i] = c2;
    }
}

// The same three numbers per row from the partials the Gram tiles left (ahc_gram_mfma2_t<true>): Np / 128 partials per row, merged by (value, index).
// Workgroup = 64 rows x 4 shares of the tile columns; the shares meet in LDS.
__global__ __launch_bounds__(256) void ahc_row_minima_parts(Ws w, const double2 *__restrict__ part_vs, const int *__restrict__ part_ix) {
    __shared__ MinAcc s_acc[3][64];
    const int lane = threadIdx.x & 63, share = threadIdx.x >> 6, i = blockIdx.x * 64 + lane, nT = w.Np / GT;
    MinAcc a;
    a.v = dinf(); a.s = dinf(); a.i = INT_MAX;
    const size_t npz = static_cast<size_t>(w.Np);
    for (int t = share; t < nT; t += 4) {
        const double2 vs = part_vs[t * npz + i];
        MinAcc o; o.v = vs.x; o.s = vs.y; o.i = part_ix[t * npz + i];
        min_merge(a, o);
    }
    if (share) s_acc[share - 1][lane] = a;
    __syncthreads();
    if (share) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) min_merge(a, s_acc[q][lane]);
    const bool any = a.i != INT_MAX && w.node[i] != kDead;
    RowSt r; r.d1 = any ? a.v : dinf(); r.nn = any ? a.i : -1; r.nnnode = any ? w.node[a.i] : -1;
    w.row[i] = r;
    w.e2[i] = any ? a.s : dinf();
}

}  // namespace

namespace fa_ahc {

void startup_transpose(hipStream_t st, const double *d_data, double *XT, int N, int Np, int d) {
    hipLaunchKernelGGL(ahc_transpose, dim3((Np + 31) / 32, (d + 31) / 32), dim3(256), 0, st, d_data, XT, N, Np, d);
}

namespace {
template <bool MINIMA> hipError_t gram2_attr() {   // once per process: the two LDS stages of a tile are 72 KB of dynamic LDS
    static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_gram_mfma2_t<MINIMA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    static_cast<int>(kGram2LdsBytes));
    return e;
}
}  // namespace

// The start-up of the filter-based rounds on `st`: initial state / rows, the slot-major transpose, the matrix (AUTO: Gram form on the fp64 matrix cores,
// whose tiles leave per-tile row minima; EXACT: the reference's sequential sums), row minima, eps.  w.C holds the points already.
void startup_filter(hipStream_t st, const Ws &w, const Layout &L, char *base, int dev_mode, const double *d_data, size_t N, size_t Np, size_t d) {
    bool minima_done = false;   // the Gram tiles left per-tile row minima (ahc_gram_mfma2_t<true>): no second pass over the matrix
    hipLaunchKernelGGL(ahc_init_state, dim3(1), dim3(64), 0, st, w, dev_mode);
    hipLaunchKernelGGL(ahc_init_rows, dim3(static_cast<unsigned>((std::max(Np, 2 * N) + 255) / 256)), dim3(256), 0, st, w);
    startup_transpose(st, d_data, w.XT, w.N, w.Np, w.d);
    if (dev_mode == FA_AHC_MODE_AUTO) {  // Gram form on the fp64 matrix cores (approximate entries, see ahc_gram_mfma)
        double *d_norms = reinterpret_cast<double *>(base + L.norms);
        hipLaunchKernelGGL(ahc_sqnorms, dim3((w.Np + 63) / 64), dim3(256), 0, st, w, d_norms);
        if (w.d % G2K == 0 && !fa::sw_on(fa::Sw::AHC_GRAM_V1) && gram2_attr<true>() == hipSuccess) {
            double2 *part_vs = reinterpret_cast<double2 *>(base + L.part_vs);
            int *part_ix = reinterpret_cast<int *>(base + L.part_ix);
            hipLaunchKernelGGL(ahc_gram_mfma2_t<true>, FA_GRAM_GRID(w.Np / GT), dim3(256), kGram2LdsBytes, st, w, d_norms, part_vs, part_ix);
            hipLaunchKernelGGL(ahc_row_minima_parts, dim3(w.Np / 64), dim3(256), 0, st, w, part_vs, part_ix);
            minima_done = true;
        } else
            hipLaunchKernelGGL(ahc_gram_mfma, dim3(w.Np / GT, w.Np / GT), dim3(256), 0, st, w, d_norms);
    } else {
        const int tiles = w.Np / PT;
        hipLaunchKernelGGL(ahc_pairwise, dim3(tiles, tiles), dim3(256), 0, st, w);
    }
    if (!minima_done) hipLaunchKernelGGL(ahc_row_minima, dim3(w.Np), dim3(kBlk), 0, st, w);
    hipLaunchKernelGGL(ahc_set_eps, dim3(1), dim3(64), 0, st, w);
}

// Norms + the Gram-form matrix of `gw` (no row minima): the start-up of the matrix-filtered reference-order run, which takes its own minima.
fa_status startup_gram(fa_ctx *ctx, hipStream_t st, const Ws &gw, double *d_norms) {
    hipLaunchKernelGGL(ahc_sqnorms, dim3((gw.Np + 63) / 64), dim3(256), 0, st, gw, d_norms);
    if (gw.d % G2K == 0) {
        FA_HIP_TRY(ctx, gram2_attr<false>());
        hipLaunchKernelGGL(ahc_gram_mfma2_t<false>, FA_GRAM_GRID(gw.Np / GT), dim3(256), kGram2LdsBytes, st, gw, d_norms, static_cast<double2 *>(nullptr), static_cast<int *>(nullptr));
    } else
        hipLaunchKernelGGL(ahc_gram_mfma, dim3(gw.Np / GT, gw.Np / GT), dim3(256), 0, st, gw, d_norms);
    return FA_SUCCESS;
}

}  // namespace fa_ahc
