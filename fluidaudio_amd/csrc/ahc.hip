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
#include <algorithm>
#include <climits>
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

#include <thread>

#include "fa_common.h"
#include "ahc_reforder.h"

namespace {

#ifndef FA_AHC_SPECULATE
#define FA_AHC_SPECULATE 1   // the operands of the presumptive merge are requested before the decision is complete (0: after it, as in round 2)
#endif
#ifndef FA_AHC_BLK
#define FA_AHC_BLK 256
#endif
constexpr int kBlk = FA_AHC_BLK;   // rows per block record == threads per round workgroup (512 measured: see profiles/r03_ahc_variants.txt)
constexpr int kWaves = kBlk / 64;
constexpr int kMaxBlocks = 768;    // N <= 196 608 (N^2 * 8 B = 288 GB is reached at N ~ 190 000)
constexpr int kRoundsPerGraph = 512;  // multiple of 4 (counter rotation) and of 2 (parity)
constexpr int kMaxCand = 64;       // candidate rows inside an ambiguity window
constexpr int kMaxPairs = 1024;    // matrix entries inside an ambiguity window
constexpr int kDead = INT_MAX;     // node id of an empty slot

enum { OP_NONE = 0, OP_MERGE = 1, OP_RESCAN = 2, OP_COLLECT = 3, OP_PAIRS = 4 };

// 0 since round 3.  Round 2 went 3 -> 1 when the rows got a bound on their second minimum (e2): stale rows became rare (0 forced
// re-scans on the benchmark distributions; 9.1 -> 8.5 us per round).  Measured this round with 1: 8 .. 72 piggy-backed re-scans in
// 50 000 merges — and every piggy-backed row costs a reduction chain in both reductions of a round plus the stale-bound quantity
// and the choice logic.  With 0 a stale row is re-scanned when its bound reaches the global minimum (a forced round, still 0 of them
// on all three benchmark inputs): 7.13 -> 6.32 us per round (8 h session), 7.27 -> 6.41 (50k iid).
#ifndef FA_AHC_PIGGY
#define FA_AHC_PIGGY 0
#endif
// (Round 3 measured and rejected requesting the likely partner rows of the NEXT merge one round ahead — the partner was among 3 requested
// rows in 72 % / 59 % of the merges, yet the round got slower, 6.14 -> 6.61 us: DESIGN.md 3.3.1b, profiles/r03_ahc_variants.txt.  The
// switch FA_AHC_PREFETCH and its code left the tree in round 4.)
constexpr int kPiggy = FA_AHC_PIGGY;   // stale rows re-scanned on top of every merge / forced re-scan round
constexpr int kPend = 1 + kPiggy;  // rows whose per-block partial minima one round can produce

// Device state, double buffered by round parity; written by workgroup 0 only.  AhcHot is what EVERY thread of every round needs: it
// is fetched with a handful of 16-byte VECTOR loads issued next to the record loads (round 2 read the whole state through the
// scalar cache: with ~40 SGPRs of workspace pointers live the compiler spilled and chained the reads into three dependent scalar
// round trips that had to finish before the first record load could even be issued).  The rest is touched by one thread per round.
struct AhcHot {
    int32_t step, done, halt, need_exact, error, mode;
    int32_t prev_op;              // what the previous round executed
    int32_t sym_limit;            // nodes below this id existed when the matrix was last built in full: BOTH copies of their pairs are valid
    int32_t pend_row[kPend], pend_node[kPend];  // rows whose block-partial minima the previous round produced (-1: none)
    double eps, lim;              // lim: window limit carried COLLECT -> PAIRS -> evaluation
    int32_t n_points;             // N of THIS problem: the uniform-layout batch (ahc_round_uni) shares every other shape constant between its problems
    int32_t rounds32;             // rounds executed: carried in the hot state (the rare counters — re-scans, windows — are atomics on state[0])
};
struct alignas(16) AhcState : AhcHot {
    unsigned long long dmax_bits, nmax_bits;  // largest matrix entry / largest squared norm seen by the start-up kernels
    long long rounds, rescans, windows, piggy;
};
static_assert(sizeof(AhcState) % 16 == 0 && sizeof(AhcHot) % 8 == 0, "the state is read in 16-byte pieces");
constexpr int kHotVec = (sizeof(AhcHot) + 15) / 16;

struct WinCounters {  // 4 copies rotating with the round index: [t&3] written, [(t-1)&3] read, [(t+1)&3] cleared
    unsigned long long stale_key;  // (slot << 32 | node) of the lowest stale row inside the window
    int32_t ncand, npairs;
};

// Block records, double buffered by round parity and stored field-by-field ([2][nblk] arrays of 16-byte elements) so
// that lane i of a wave reads element i: every record load is one fully coalesced dwordx4.
struct __attribute__((aligned(16))) RecA { double v1; int cnt, pad; };        // smallest row minimum of the block (bounds of
                                                                              // stale rows included); rows within 2 eps of it
struct __attribute__((aligned(16))) RecS { double sv; int srow, snode; };     // smallest bound among the block's stale rows
struct __attribute__((aligned(16))) RecP { double pv; int slot, node; };      // block-partial minimum of a row being produced
// recI: int4 {r1, q1, node(r1), node(q1)}: the row holding v1, its neighbour slot (-1: stale), their node ids

struct __attribute__((aligned(16))) RowSt {  // per slot, owned by thread (slot & 255) of workgroup (slot >> 8)
    double d1;               // minimum over all other live slots (lower bound while nn < 0)
    int nn, nnnode;          // nearest neighbour slot (lowest on ties; -1: merged away) and its node id
};

struct Ws {   // what a round touches first comes first: with kernel-argument preloading (Makefile: -amdgpu-kernarg-preload-count) the leading
              // 16 dwords arrive in SGPRs with the wavefront instead of through a scalar load at its start
    int32_t nblk, Np;
    AhcState *state;   // [2]
    RecA *recA;      // [2][nblk]
    int4 *recI;      // [2][nblk]
    RecP *recP;      // [2][kPend][nblk]
    RowSt *row;      // [Np]
    int32_t *node;   // [Np]
    double *e2;      // [Np]   lower bound of the row's entries OTHER than the nearest neighbour's (see the row update of the round)
    int32_t N, d;
    int32_t *flags;    // [0]: a NaN distance was seen (nan_error, FastClusterWrapper.cpp:60-62)
    double *M;       // [Np][Np]
    double *C;       // [2N][d]  centroids by node id (rows 0..N-1 = input points)
    double *XT;      // [d][Np]  slot-major transposed coordinates (init; maintained in EXACT mode only)
    double *sizes;   // [2N]     cluster size by node id
    double *Z;       // [(N-1)*4]
    RecS *recS;      // [2][nblk]
    int2 *cand;      // [kMaxCand]  slot, node
    int4 *pairs;     // [kMaxPairs] a, b, node a, node b
    WinCounters *cnt;  // [4]
    unsigned long long *prof;  // [16] cycle counters (FA_AHC_PROFILE builds only)
};

__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }
__device__ __forceinline__ bool lt2(double v, int i, double ov, int oi) { return v < ov || (v == ov && i < oi); }

// Matrix entry of the pair (row slot r holding node nr, column slot x holding node nx), read by the thread that owns column x.
// The copy in row r is valid when r holds the younger node (a merge rewrites exactly that row) — and also when BOTH nodes already existed
// at the last full build of the matrix (start-up, exact rebuild), which wrote both copies: taking the row copy then keeps the access
// coalesced across the wavefront.  Without the second case the pairs of a single point r with the ~N/2 points of higher index were read
// as M[x][r]: one 8-byte load per lane, each in a different 400 KB row (a different page) — the bulk of a round's memory time.
__device__ __forceinline__ double pair_entry(const double *M, const int Np, const int r, const int nr, const int x, const int nx, const int sym_limit) {
    const bool row_copy = nr > nx || (nr < sym_limit && nx < sym_limit);
    return row_copy ? M[static_cast<size_t>(r) * Np + x] : M[static_cast<size_t>(x) * Np + r];
}

// ------------------------------------------------------------------------------ wave helpers (DPP)
// A 64-lane reduction through __shfl_xor costs ~6 dependent ds_bpermute round trips per 32-bit word (measured
// ~1000 cycles per step for the 8-word payloads this kernel needs); DPP row shifts + row broadcasts stay in the
// VALU.  Values are non-negative doubles (or +inf), never NaN.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(const double old, const double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_lane63(const double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_umin(const unsigned v) {
    const unsigned o = static_cast<unsigned>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), CTRL, ROWMASK, 0xf, false));
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned wave_umin(unsigned v) {  // result uniform
    v = dpp_umin<0x111, 0xf>(v);  // row_shr:1
    v = dpp_umin<0x112, 0xf>(v);  // row_shr:2
    v = dpp_umin<0x114, 0xf>(v);  // row_shr:4
    v = dpp_umin<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row minimum
    v = dpp_umin<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
    v = dpp_umin<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3 -> lane 63 holds the minimum
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
// Minimum of non-negative doubles (+0 .. +inf: the IEEE bit pattern is monotone) as two 32-bit DPP reductions;
// v_min_u32 takes the DPP operand directly, v_min_f64 would need two v_mov_dpp per step.
__device__ __forceinline__ double wave_min(const double v) {  // result uniform
    const unsigned hi = static_cast<unsigned>(__double2hiint(v)), lo = static_cast<unsigned>(__double2loint(v));
    const unsigned mhi = wave_umin(hi);
    const unsigned mlo = wave_umin(hi == mhi ? lo : 0xffffffffu);
    return __hiloint2double(static_cast<int>(mhi), static_cast<int>(mlo));
}
// NQ independent minimum reductions advanced in lock step: the DPP chains interleave, so no wait states are spent
// between dependent steps (a single chain needs 2 idle slots after every VALU write that a DPP read consumes).
template <int NQ>
__device__ __forceinline__ void wave_umin_multi(unsigned (&v)[NQ]) {
#define FA_AHC_STEP(CTRL, MASK) _Pragma("unroll") for (int q = 0; q < NQ; ++q) v[q] = dpp_umin<CTRL, MASK>(v[q]);
    FA_AHC_STEP(0x111, 0xf) FA_AHC_STEP(0x112, 0xf) FA_AHC_STEP(0x114, 0xf) FA_AHC_STEP(0x118, 0xf)
    FA_AHC_STEP(0x142, 0xa) FA_AHC_STEP(0x143, 0xc)
#undef FA_AHC_STEP
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v[q]), 63));
}
// minima m[q] of NQ non-negative doubles per lane and the lowest lane L[q] holding each (uniform results)
template <int NQ>
__device__ __forceinline__ void wave_min_multi(const double (&key)[NQ], double (&m)[NQ], int (&L)[NQ]) {
    unsigned hi[NQ], lo[NQ], mh[NQ], ml[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { hi[q] = static_cast<unsigned>(__double2hiint(key[q])); lo[q] = static_cast<unsigned>(__double2loint(key[q])); mh[q] = hi[q]; }
    wave_umin_multi<NQ>(mh);
#pragma unroll
    for (int q = 0; q < NQ; ++q) ml[q] = hi[q] == mh[q] ? lo[q] : 0xffffffffu;
    wave_umin_multi<NQ>(ml);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        m[q] = __hiloint2double(static_cast<int>(mh[q]), static_cast<int>(ml[q]));
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hi[q] == mh[q] && lo[q] == ml[q]);
        L[q] = __builtin_amdgcn_readfirstlane(mask ? __ffsll(static_cast<long long>(mask)) - 1 : 0);
    }
}
// Workgroup barrier for an exchange through LDS: the LDS writes of this wave are complete (lgkmcnt), nothing is said about global memory.
// __syncthreads() is a workgroup-scope release + acquire: on gfx950 it also waits (vmcnt(0)) until every global STORE issued so far has
// been acknowledged — in the round kernel the rewritten matrix row and the row states, i.e. a memory round trip in front of the block record.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// number of lanes whose predicate holds (uniform)
__device__ __forceinline__ int wave_count(const bool p) { return __popcll(__builtin_amdgcn_ballot_w64(p)); }

__device__ __forceinline__ double wave_sum(double v) {  // fixed association order; result uniform
    v += dpp_f64<0x111, 0xf>(0.0, v);
    v += dpp_f64<0x112, 0xf>(0.0, v);
    v += dpp_f64<0x114, 0xf>(0.0, v);
    v += dpp_f64<0x118, 0xf>(0.0, v);
    v += dpp_f64<0x142, 0xa>(0.0, v);
    v += dpp_f64<0x143, 0xc>(0.0, v);
    return bcast_lane63(v);
}
// lowest lane whose value equals the (uniform) minimum
__device__ __forceinline__ int first_lane_eq(const double v, const double m) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(v == m);
    return __builtin_amdgcn_readfirstlane(mask ? __ffsll(static_cast<long long>(mask)) - 1 : 0);
}
__device__ __forceinline__ int lane_value(const int v, const int lane) { return __builtin_amdgcn_readlane(v, lane); }

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
constexpr int GT = 128, GK = 16, GS = 144;

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

template <bool MINIMA>
__global__ __launch_bounds__(256, 2) void ahc_gram_mfma2_t(Ws w, const double *__restrict__ norms, double2 *__restrict__ part_vs, int *__restrict__ part_ix) {
    extern __shared__ __attribute__((aligned(16))) double sg[];
    if (blockIdx.x > blockIdx.y) return;   // symmetric: tiles on and below the diagonal, off-diagonal tiles are written twice
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;   // wave: in an SGPR, so the row addresses below are scalar
    const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
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
            const size_t at = blockIdx.x * npz + i0 + wr + lane;
            part_vs[at] = make_double2(mine.v, mine.s);
            part_ix[at] = mine.i;
        }
        if (!wr && mirror) {
            min_merge(cmine, reinterpret_cast<const MinAcc *>(sg + (wave + 2) * 64 * TS)[64 + lane]);    // wave (64, wc) = this wave + 2
            const size_t at = blockIdx.y * npz + j0 + wc + lane;
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
#pragma unroll 8
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

// ------------------------------------------------------------------------------ block record
// Per workgroup, for the NEXT round: the smallest row minimum (+ its row), how many rows lie within 2 eps of it, the
// smallest stale bound, and the block-partial minima of the rows being produced.  Per wave: 2 + kPend interleaved
// DPP min-reductions (winner lanes by ballot); across the four waves: LDS + ONE __syncthreads; thread 0 writes.
constexpr int kStaleQ = kPiggy > 0 ? 1 : 0;   // the stale-bound quantity only feeds the choice of piggy-backed rows
constexpr int kNQ = 1 + kStaleQ + kPend;
struct __attribute__((aligned(8))) QOut { double v; int a, b, c, d; };   // one reduced quantity of one wave: value + payload
struct WaveOut {                                                          // quantity 0: smallest row minimum (r1, q1, node r1, node q1)
    QOut q[kWaves][kNQ];                                                  //          1: smallest stale bound (row, node)
    int cnt[kWaves];                                                      //      2 + p: partial minimum of produced row p (slot, node)
};

// ---- a thread's CPT consecutive slots (columns x0 .. x0 + CPT - 1; x0 is a multiple of CPT, the arrays are 256-byte aligned): one request per
// array and thread instead of CPT.  CPT = 1 is the round-2 .. 4 form (one slot per thread).
template <int CPT> __device__ __forceinline__ void load_i32(const int *p, int (&v)[CPT]) {
    static_assert(CPT == 1 || CPT == 2 || CPT == 4, "columns per thread");
    if constexpr (CPT == 4) { const int4 q = *reinterpret_cast<const int4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else if constexpr (CPT == 2) { const int2 q = *reinterpret_cast<const int2 *>(p); v[0] = q.x; v[1] = q.y; }
    else v[0] = p[0];
}
template <int CPT> __device__ __forceinline__ void load_f64(const double *p, double (&v)[CPT]) {
    if constexpr (CPT == 1) v[0] = p[0];
    else {
#pragma unroll
        for (int j = 0; j < CPT; j += 2) { const double2 q = *reinterpret_cast<const double2 *>(p + j); v[j] = q.x; v[j + 1] = q.y; }
    }
}
template <int CPT> __device__ __forceinline__ void store_f64(double *p, const double (&v)[CPT]) {
    if constexpr (CPT == 1) p[0] = v[0];
    else {
#pragma unroll
        for (int j = 0; j < CPT; j += 2) *reinterpret_cast<double2 *>(p + j) = make_double2(v[j], v[j + 1]);
    }
}
// pair_entry for the thread's CPT columns against row slot r (node nr); columns that are dead or equal skip0 / skip1 get 0.  The ROW copies of the
// CPT columns are one contiguous piece of row r and are requested together whenever any column wants an entry (the bytes share cache lines with
// the wanted ones); a column whose valid copy is the column copy M[x][r] is requested on top — the rare case (see pair_entry).
template <int CPT>
__device__ __forceinline__ void pair_entries(const double *M, const int Np, const int r, const int nr, const int x0, const int (&nx)[CPT], const int sym_limit,
                                             const int skip0, const int skip1, double (&out)[CPT]) {
    if constexpr (CPT == 1) {
        out[0] = (nx[0] != kDead && x0 != skip0 && x0 != skip1) ? pair_entry(M, Np, r, nr, x0, nx[0], sym_limit) : 0.0;
    } else {
        // No branch per column: every column requests ONE entry from an address that is always valid — its column copy where that is the valid one,
        // else its row copy (also for a column that wants nothing: the value is dropped).  CPT independent requests, issued back to back.
        const double *rowp = M + static_cast<size_t>(r) * Np + x0;
        double v[CPT];
        bool want[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            want[j] = nx[j] != kDead && x0 + j != skip0 && x0 + j != skip1;
            const bool rc = nr > nx[j] || (nr < sym_limit && nx[j] < sym_limit);
            const double *colp = M + static_cast<size_t>(x0 + j) * Np + r;
            const double *pj = (want[j] && !rc) ? colp : rowp + j;
            v[j] = *pj;
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) out[j] = want[j] ? v[j] : 0.0;
    }
}

// `key` is the thread's smallest row minimum over its CPT slots (x, nx, nnx, nnnodex: that slot's), `keys_all` all of them (the window count).
template <int CPT>
__device__ __forceinline__ void block_record(const Ws &w, const int par, const int blk, const double eps, const double key, const double (&keys_all)[CPT],
                                             const double skey, const double (&pkey)[kPend], const int (&pslot)[kPend],
                                             const int (&pnode)[kPend], const int x, const int nx, const int nnx,
                                             const int nnnodex, WaveOut *s_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double keys[kNQ], m[kNQ];
    int L[kNQ];
    constexpr int kP0 = 1 + kStaleQ;   // first produced-row quantity
    keys[0] = key;
    if (kStaleQ) keys[kStaleQ] = skey;
#pragma unroll
    for (int p = 0; p < kPend; ++p) keys[kP0 + p] = pkey[p];
    wave_min_multi<kNQ>(keys, m, L);
    QOut o[kNQ];
    o[0].v = m[0]; o[0].a = lane_value(x, L[0]); o[0].b = lane_value(nnx, L[0]); o[0].c = lane_value(nx, L[0]); o[0].d = lane_value(nnnodex, L[0]);
    if (kStaleQ) { o[kStaleQ].v = m[kStaleQ]; o[kStaleQ].a = lane_value(x, L[kStaleQ]); o[kStaleQ].b = lane_value(nx, L[kStaleQ]); o[kStaleQ].c = 0; o[kStaleQ].d = 0; }
#pragma unroll
    for (int p = 0; p < kPend; ++p) { o[kP0 + p].v = m[kP0 + p]; o[kP0 + p].a = lane_value(pslot[p], L[kP0 + p]); o[kP0 + p].b = lane_value(pnode[p], L[kP0 + p]); o[kP0 + p].c = 0; o[kP0 + p].d = 0; }
    // rows of this wave within 2 eps of its minimum: exact up to 3 per lane — the decision only asks whether the window holds exactly two
    const double wl = m[0] + 2.0 * eps;
    int mine = 0;
#pragma unroll
    for (int j = 0; j < CPT; ++j) mine += (keys_all[j] <= wl && keys_all[j] < dinf()) ? 1 : 0;
    int cnt = wave_count(mine >= 1);
    if (CPT >= 2) cnt += wave_count(mine >= 2);
    if (CPT >= 3) cnt += wave_count(mine >= 3);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kNQ; ++q) s_out->q[wave][q] = o[q];
        s_out->cnt[wave] = cnt;
    }
    lds_barrier();   // LDS exchange only: __syncthreads() would also wait for the row / matrix stores of this round to reach memory (a store round trip on the critical path)
    if (tid >= kNQ) return;
    // lane q of wave 0 merges quantity q of the four waves (ties -> lowest wave == lowest rows) and writes its own record:
    // six short chains side by side instead of one thread walking all six
    QOut best = s_out->q[0][tid];
    QOut other[kWaves];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) other[wv] = s_out->q[wv][tid];
    const double v0 = best.v;
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) if (other[wv].v < best.v) best = other[wv];
    const size_t o1 = static_cast<size_t>(par) * w.nblk + blk;
    if (tid == 0) {
        RecA ra; ra.v1 = best.v; ra.cnt = 0; ra.pad = 0;
        // rows within 2 eps of the block minimum, counted conservatively (a wave's rows were counted against ITS minimum)
        if (v0 <= best.v + 2.0 * eps) ra.cnt += s_out->cnt[0];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) if (other[wv].v <= best.v + 2.0 * eps) ra.cnt += s_out->cnt[wv];
        const bool any = best.v < dinf();
        w.recA[o1] = ra;
        w.recI[o1] = any ? make_int4(best.a, best.b, best.c, best.d) : make_int4(-1, -1, -1, -1);
    } else if (kStaleQ && tid == 1) {
        RecS rsv; rsv.sv = best.v;
        const bool any = best.v < dinf();
        rsv.srow = any ? best.a : -1; rsv.snode = any ? best.b : -1;
        w.recS[o1] = rsv;
    } else {
        RecP rp; rp.pv = best.v;
        const bool any = best.v < dinf();
        rp.slot = any ? best.a : -1; rp.node = any ? best.b : -1;
        w.recP[(static_cast<size_t>(par) * kPend + (tid - kP0)) * w.nblk + blk] = rp;
    }
}

template <int CPT>
__global__ __launch_bounds__(kBlk) void ahc_records(Ws w) {  // records of BOTH parities (blockIdx.y) from the row arrays; one workgroup per block of kBlk * CPT slots.
    // Parity 1 too: a round requests the operands of its presumptive merge from the records BEFORE it looks at the halt flag, and a run that halts in
    // round 0 (a NaN met by the start-up) has never written parity 1 — round 1 then formed addresses from whatever the workspace held there (harmless
    // while that was zeros or an older run's records; after a reference-order run had used the same bytes: a memory fault, found by the tests of round 5).
    __shared__ WaveOut s_out[1];
    const int tid = threadIdx.x, blk = blockIdx.x, x0 = (blk * kBlk + tid) * CPT;
    int nx[CPT];
    load_i32<CPT>(w.node + x0, nx);
    double pkey[kPend];
    int pslot[kPend], pnode[kPend];
#pragma unroll
    for (int p = 0; p < kPend; ++p) { pkey[p] = dinf(); pslot[p] = -1; pnode[p] = -1; }
    double keys_all[CPT], key = dinf(), skey = dinf();
    int bx = x0, bnx = nx[0], bnn = -1, bnnnode = -1;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const RowSt r = w.row[x0 + j];
        const bool live = nx[j] != kDead;
        keys_all[j] = live ? r.d1 : dinf();
        if (j == 0 || keys_all[j] < key) { key = keys_all[j]; bx = x0 + j; bnx = nx[j]; bnn = r.nn; bnnnode = r.nnnode; }
        if (live && r.nn < 0 && r.d1 < skey) skey = r.d1;
    }
    block_record<CPT>(w, static_cast<int>(blockIdx.y), blk, w.state[0].eps, key, keys_all, skey, pkey, pslot, pnode, bx, bnx, bnn, bnnnode, s_out);
}

// ------------------------------------------------------------------------------ the round kernel
struct Decision {
    int op;
    int a, b, na, nb;  // MERGE: slots (a < b) and their node ids; RESCAN: a = row, na = its node
    double dab;        // exact distance when known (window evaluation), else < 0
    double lim;
    int halt, need_exact, error, done;
};

// Exact squared distances of the listed pairs, the reference's summation order (sequential in k, one rounding per
// operation; FastClusterWrapper.cpp:68-75).  One wavefront per pair: 64 lanes square the differences of a 64-wide
// slice, lane 0 adds them in index order.  Minimum by (value, a, b); returned in every thread.
__device__ void exact_min_pair(const Ws &w, const int np, double *s_sq /*[kWaves*64]*/, double *s_val, int *s_idx,
                               double &best, int &best_p, bool &tie) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = w.d;
    best = dinf();
    best_p = INT_MAX;
    bool nan_seen = false, tie_w = false;   // tie: two DIFFERENT pairs share the exact minimum (the same pair may be listed twice, once from each of its rows)
    for (int p0 = 0; p0 < np; p0 += kWaves) {
        const int p = p0 + wave;
        const bool live = p < np;
        const int4 pr = live ? w.pairs[p] : make_int4(0, 0, 0, 0);
        const double *ca = w.C + static_cast<size_t>(pr.z) * d, *cb = w.C + static_cast<size_t>(pr.w) * d;
        double sum = 0.0;
        for (int k0 = 0; k0 < d; k0 += 64) {
            const int k = k0 + lane;
            double sq = 0.0;
            if (live && k < d) { const double diff = __dsub_rn(ca[k], cb[k]); sq = __dmul_rn(diff, diff); }
            s_sq[wave * 64 + lane] = sq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0 && live) {
                const int n = d - k0 < 64 ? d - k0 : 64;
                for (int j = 0; j < n; ++j) sum = __dadd_rn(sum, s_sq[wave * 64 + j]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0 && live) {
            if (sum != sum) nan_seen = true;
            else if (best_p == INT_MAX || sum < best) { best = sum; best_p = p; tie_w = false; }
            else if (sum == best) {
                const int4 bp = w.pairs[best_p];
                if (pr.x != bp.x || pr.y != bp.y) tie_w = true;
                if (pr.x < bp.x || (pr.x == bp.x && pr.y < bp.y)) best_p = p;
            }
        }
    }
    if (lane == 0) { s_val[wave] = nan_seen ? -1.0 : best; s_idx[wave] = best_p; s_idx[kWaves + wave] = tie_w ? 1 : 0; }
    __syncthreads();
    best = dinf(); best_p = INT_MAX;
    tie = false;
    bool bad = false;
    for (int wv = 0; wv < kWaves; ++wv) {
        const double v = s_val[wv];
        const int p = s_idx[wv];
        if (v < 0.0) bad = true;
        if (p == INT_MAX) continue;
        bool take = best_p == INT_MAX || v < best;
        if (take) tie = s_idx[kWaves + wv] != 0;
        else if (v == best) {
            const int4 q = w.pairs[p], bq = w.pairs[best_p];
            if (q.x != bq.x || q.y != bq.y || s_idx[kWaves + wv] != 0) tie = true;
            take = q.x < bq.x || (q.x == bq.x && q.y < bq.y);
        }
        if (take) { best = v; best_p = p; }
    }
    __syncthreads();
    if (bad) best_p = -1;  // NaN distance -> nan_error in the reference
}

#ifdef FA_AHC_PROFILE   // 1: every stamp waits for all outstanding memory operations (phase costs in isolation); 2: stamps only (the overlapped timeline)
#define AHC_STAMP(i)                                                                  \
    do {                                                                              \
        if (FA_AHC_PROFILE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long t_now = clock64();                                   \
        t_seg[i] = t_now - t_prev;                                                    \
        t_prev = t_now;                                                               \
    } while (0)
#else
#define AHC_STAMP(i) do {} while (0)
#endif

// Phase 1 of a round: the block records of the previous round -> one decision, identical in every workgroup.  Each of the four waves
// reduces ALL block records for its own share of the QUANTITIES (lane l owns the blocks [l c, (l + 1) c): lane order == row order, so
// ballot + ffs breaks ties towards the lowest row): the results need no cross-wave merge — round 1/2 had every wave reduce a
// quarter of the blocks for all six quantities and every thread then merged four partial results (~200 dependent instructions).
//   wave 0: smallest row minimum (+ its row, neighbour, node ids) and the rows within 2 eps of it
//   wave 1: the smallest stale bound of each QUARTER of the blocks (candidates for the piggy-backed re-scans)
//   wave 2: partial minima of the produced rows 1, 2        wave 3: of the produced rows 0, 3
constexpr int kMaxC = (kMaxBlocks + 63) / 64;  // block records per lane
// A record is ONE 16-byte load.  (Round 2 read the records as structs inside a `for (j < kMaxC) { if (!(j < c)) continue; ... }` loop: the
// compiler split every struct into a value load and a payload load that it issued only after comparing the value, and chained the
// iterations — for 50 000 points 4 (wave 0) to 8 (the produced-row wave) DEPENDENT L2 / MALL round trips at the start of every round
// instead of one.  Now all records of a lane are requested before the first one is looked at: straight-line code, one case per count.)
__device__ __forceinline__ int4 rec16(const void *base, const size_t idx) { return reinterpret_cast<const int4 *>(base)[idx]; }
__device__ __forceinline__ double rec_f64(const int4 r) { return __hiloint2double(r.y, r.x); }
struct Dec {
    double v1, sv[kWaves], pd[kPend];
    int cnt, r1, q1, nr1, nq1, pad0, pad1, pad2;
    int srow[kWaves], snode[kWaves], ps[kPend], pn[kPend];
};

// BATCH: the same round for several independent problems at once (fa_ahc_linkage_batch): workgroup b works on problem
// blkmap[b].x as its block blkmap[b].y; the problem's workspace descriptor comes from a table in HBM (written before the
// first launch, constant afterwards: read through the constant address space, i.e. with scalar loads, like a kernel argument).
// the round itself; `w` = the problem's workspace, `blk` = this workgroup's block of 256 slots (see the three entry kernels below)
// N_IN_STATE: the point count comes from the problem's state (uniform-layout batch: every other shape constant is shared by its problems).
// Its arrays that are first touched AFTER the round's first batch of requests (matrix, centroids, sizes, dendrogram, window buffers) arrive
// as problem 0's and are moved by `late_shift` bytes behind that batch: their pointers come out of scalar loads of the argument segment, and an
// addition in front of the first request would put the wait for those loads there.
// BIG: more than 65 536 points, i.e. more than four block records per lane in the first reduction.  That path holds 2 x 12 records in registers
// and alone raised the whole kernel from 106 to 180 VGPRs (2 instead of 4 wavefronts per SIMD): it is compiled only into the kernels that
// serve such problems, so that four times as many workgroups of the common sizes are resident — what a launch over several problems needs.
// CPT: slots (columns) per thread.  A block = kBlk * CPT consecutive slots, thread t owns the CPT consecutive slots from (blk * kBlk + t) * CPT on (lane
// order == row order as before).  1 is the latency-optimal form of the single chain (the fewest dependent instructions per round).  A launch
// over several problems (ahc_round_uni) is bound by instruction ISSUE instead — every workgroup repeats the reduction of all block records, every
// wavefront its DPP reductions, the centroid sum, the decision arithmetic: ~830 instructions per wavefront and round whatever it owns — so
// there a thread owns 4 slots: a quarter of the workgroups, wavefronts and block records per problem, and only the per-slot part of a round
// (the two matrix entries, the Lance-Williams value, the row bookkeeping) is repeated per slot.
template <bool N_IN_STATE = false, bool BIG = true, int CPT = 1>
__device__ __forceinline__ void ahc_round_body(const Ws w_in, const int blk, const int ph /* round index & 3 */, const size_t late_shift = 0) {
    static_assert(CPT == 1 || kPiggy == 0, "piggy-backed re-scans were only ever built for one slot per thread");
    constexpr int kCols = kBlk * CPT;
    Ws w = w_in;
    extern __shared__ double s_cvec[];  // [d] merged centroid (EXACT rows)
    __shared__ WaveOut s_out[1];
    __shared__ Dec s_dec;
    __shared__ double s_sq[kBlk];
    __shared__ double s_val[kWaves];
    __shared__ int s_idx[2 * kWaves];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x0 = (blk * kBlk + tid) * CPT;   // x0: the first of this thread's slots
    const int par = ph & 1, npar = par ^ 1;
    const int Np = w.Np, nblk = w.nblk, d = w.d, N_arg = w.N;
#ifdef FA_AHC_PROFILE
    const int prof_blk = nblk / 2;
    unsigned long long t_seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = clock64();
#endif
    // Every load of the round's first memory round trip is requested before anything is waited for or branched on: the block records
    // this wave reduces (their addresses depend on kernel arguments only), the hot part of the state, the own row state.
    const int c = (nblk + 63) >> 6;
    const size_t ro = static_cast<size_t>(par) * nblk;
    const bool row_wave = wave >= 2 && (wave == 2 ? 1 : 0) < kPend;      // waves that finish produced rows: 3 (row 0 [, 3]) and, with piggy-backed rows, 2 (rows 1, 2)
    const int pk0 = wave == 2 ? 1 : 0, pk1r = wave == 2 ? 2 : 3;
    const bool phas1 = pk1r < kPend;
    const int pk1 = phas1 ? pk1r : pk0;
    constexpr int kC4 = 4 / CPT;                                        // block records per lane held in registers (N <= 65 536 = 64 lanes x kC4 x kBlk x CPT); beyond: the generic path
    int4 q0[kC4], q1[kC4];
    bool qok[kC4];
    {
        const void *b0 = wave == 0 ? static_cast<const void *>(w.recA + ro) : static_cast<const void *>(w.recP + (static_cast<size_t>(par) * kPend + pk0) * nblk);
        const void *b1 = wave == 0 ? static_cast<const void *>(w.recI + ro) : static_cast<const void *>(w.recP + (static_cast<size_t>(par) * kPend + pk1) * nblk);
#pragma unroll
        for (int j = 0; j < kC4; ++j) {
            const int i = lane * c + j;
            qok[j] = c <= kC4 && j < c && i < nblk && (wave == 0 || row_wave);
            const size_t ii = qok[j] ? i : 0;
            q0[j] = rec16(b0, ii);
            q1[j] = rec16(b1, ii);
        }
    }
    int vz = 0;
    asm volatile("" : "+v"(vz));                                        // an opaque 0 in a VGPR: keeps the state on the vector memory path
    const char *sp = reinterpret_cast<const char *>(w.state + par) + vz;
    int4 hraw[kHotVec];
#pragma unroll
    for (int i = 0; i < kHotVec; ++i) hraw[i] = reinterpret_cast<const int4 *>(sp)[i];
    // The cold part of the state (counters of rare events; the start-up maxima) is NOT read by the rounds (round 4): those counters live in
    // state[0] only and thread (0, 0) bumps them with atomic adds that nobody waits for — in the rounds where the event happens; the round
    // counter itself travels in the hot state.  Round 3 fetched the cold part in every thread next to the hot part (a broadcast line, but
    // 12 VGPRs per thread for the whole round and three more requests in the first batch).
    AhcState *const nst = w.state + npar;
    int nx[CPT];
    RowSt rs[CPT];
    double e2x[CPT];        // lower bound of the entries of a row other than its nearest neighbour's
    load_i32<CPT>(w.node + x0, nx);
#pragma unroll
    for (int j = 0; j < CPT; ++j) rs[j] = w.row[x0 + j];
    load_f64<CPT>(w.e2 + x0, e2x);
    const int nanflag = w.flags[0];
    __builtin_amdgcn_sched_barrier(0);
#ifndef FA_AHC_LATE_KERNARGS
    // Kernel arguments that are first used in the MIDDLE of the dependent chain (N: the step test and the node id of the merged cluster, Np:
    // the row addresses of the operands, cnt / pairs / cand: the rare window paths, which the compiler loads together with N): left to the
    // scheduler they are scalar loads right where they are used, i.e. two scalar-cache round trips inside the chain.  Naming them here puts
    // the loads next to the round's first memory round trip.
    asm volatile("" :: "s"(N_arg), "s"(Np), "s"(d), "s"(w.cnt), "s"(w.pairs), "s"(w.cand));
#endif
    AhcHot st;
    __builtin_memcpy(&st, hraw, sizeof(AhcHot));
    const int N = N_IN_STATE ? st.n_points : N_arg;
    if (N_IN_STATE) {
        auto at = [late_shift](auto *q) { return reinterpret_cast<decltype(q)>(reinterpret_cast<char *>(q) + late_shift); };
        w.M = at(w.M); w.C = at(w.C); w.XT = at(w.XT); w.sizes = at(w.sizes); w.Z = at(w.Z); w.recS = at(w.recS);
        w.cand = at(w.cand); w.pairs = at(w.pairs); w.cnt = at(w.cnt); w.prof = at(w.prof);
    }
    AhcHot *const nhot = nst;                                          // the next round's state: the hot 64 bytes only (the cold part stays in state[0])
    auto bump = [&](long long *counter) { atomicAdd(reinterpret_cast<unsigned long long *>(counter), 1ULL); };

    // ---- phase 1: every workgroup reduces the same records -> the same decision ------------------------------------
    const int perw = (nblk + kWaves - 1) / kWaves;
    // wave 0: the smallest row minimum over all blocks (+ its row, neighbour, node ids) and the rows within 2 eps of it
    auto reduce_minimum = [&](auto cc, const int4 *ra, const int4 *ri, const bool *ok) {
        constexpr int C = decltype(cc)::value;
        double va[C], key = dinf();
        int ca[C];
        int4 ids = make_int4(-1, -1, -1, -1);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            va[j] = ok[j] ? rec_f64(ra[j]) : dinf();
            ca[j] = ok[j] ? ra[j].z : 0;
            const bool better = va[j] < key;
            key = better ? va[j] : key;
            ids.x = better ? ri[j].x : ids.x; ids.y = better ? ri[j].y : ids.y; ids.z = better ? ri[j].z : ids.z; ids.w = better ? ri[j].w : ids.w;
        }
        AHC_STAMP(0);
        const double keys[1] = {key};
        double m[1];
        int L[1];
        wave_min_multi<1>(keys, m, L);
        int cl = 0;
        const double wl = m[0] + 2.0 * st.eps;
#pragma unroll
        for (int j = 0; j < C; ++j) cl += (va[j] <= wl && va[j] < dinf()) ? ca[j] : 0;
        const int cnt = wave_count(cl >= 1) + wave_count(cl >= 2) + wave_count(cl >= 3);  // exact up to 3 per lane; only "== 2" matters
        const int r1 = lane_value(ids.x, L[0]), q1_ = lane_value(ids.y, L[0]), nr1 = lane_value(ids.z, L[0]), nq1 = lane_value(ids.w, L[0]);
        if (lane == 0) { s_dec.v1 = m[0]; s_dec.cnt = cnt; s_dec.r1 = r1; s_dec.q1 = q1_; s_dec.nr1 = nr1; s_dec.nq1 = nq1; }
    };
    // waves 2 / 3: block-partial minima of the rows the previous round produced -> their minimum and nearest neighbour
    auto reduce_rows = [&](auto cc, const int4 *r0, const int4 *r1, const bool *ok) {
        constexpr int C = decltype(cc)::value;
        double keys[2] = {dinf(), dinf()};
        int ps[2] = {-1, -1}, pn[2] = {-1, -1};
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double v0 = ok[j] ? rec_f64(r0[j]) : dinf(), v1 = ok[j] && phas1 ? rec_f64(r1[j]) : dinf();
            const bool b0 = v0 < keys[0], b1 = v1 < keys[1];
            keys[0] = b0 ? v0 : keys[0]; ps[0] = b0 ? r0[j].z : ps[0]; pn[0] = b0 ? r0[j].w : pn[0];
            keys[1] = b1 ? v1 : keys[1]; ps[1] = b1 ? r1[j].z : ps[1]; pn[1] = b1 ? r1[j].w : pn[1];
        }
        AHC_STAMP(0);
        double m[2];
        int L[2];
        wave_min_multi<2>(keys, m, L);
        const int a0 = lane_value(ps[0], L[0]), b0 = lane_value(pn[0], L[0]), a1 = lane_value(ps[1], L[1]), b1 = lane_value(pn[1], L[1]);
        if (lane == 0) { s_dec.pd[pk0] = m[0]; s_dec.ps[pk0] = a0; s_dec.pn[pk0] = b0; if (phas1) { s_dec.pd[pk1] = m[1]; s_dec.ps[pk1] = a1; s_dec.pn[pk1] = b1; } }
    };
    if (c <= kC4) {
        if (wave == 0) reduce_minimum(std::integral_constant<int, kC4>{}, q0, q1, qok);
        else if (row_wave) reduce_rows(std::integral_constant<int, kC4>{}, q0, q1, qok);
    } else if (BIG && (wave == 0 || row_wave)) {   // more than 65 536 points: 5 .. 12 records per lane, requested together, then the same reductions
        int4 g0[kMaxC], g1[kMaxC];
        bool gok[kMaxC];
        const void *b0 = wave == 0 ? static_cast<const void *>(w.recA + ro) : static_cast<const void *>(w.recP + (static_cast<size_t>(par) * kPend + pk0) * nblk);
        const void *b1 = wave == 0 ? static_cast<const void *>(w.recI + ro) : static_cast<const void *>(w.recP + (static_cast<size_t>(par) * kPend + pk1) * nblk);
#pragma unroll
        for (int j = 0; j < kMaxC; ++j) {
            const int i = lane * c + j;
            gok[j] = j < c && i < nblk;
            const size_t ii = gok[j] ? i : 0;
            g0[j] = rec16(b0, ii);
            g1[j] = rec16(b1, ii);
        }
        if (wave == 0) reduce_minimum(std::integral_constant<int, kMaxC>{}, g0, g1, gok);
        else reduce_rows(std::integral_constant<int, kMaxC>{}, g0, g1, gok);
    }
    if (kPiggy > 0 && wave == 1) {   // the smallest stale bound of each QUARTER of the blocks (candidates for the piggy-backed re-scans)
        double keys[kWaves];
        int srow[kWaves], snode[kWaves];
#pragma unroll
        for (int q = 0; q < kWaves; ++q) { keys[q] = dinf(); srow[q] = -1; snode[q] = -1; }
#pragma unroll
        for (int j = 0; j < kMaxC; ++j) {
            if (j > 0 && !(j < c)) continue;
            const int i = lane * c + j;
            const bool ok = j < c && i < nblk;
            const RecS rv = w.recS[ro + (ok ? i : 0)];
            if (!ok) continue;
            const int q = (i >= perw ? 1 : 0) + (i >= 2 * perw ? 1 : 0) + (i >= 3 * perw ? 1 : 0);   // i / perw (< kWaves) without the ~40-instruction integer division
#pragma unroll
            for (int qq = 0; qq < kWaves; ++qq)
                if (qq == q && rv.sv < keys[qq]) { keys[qq] = rv.sv; srow[qq] = rv.srow; snode[qq] = rv.snode; }
        }
        AHC_STAMP(0);
        double m[kWaves];
        int L[kWaves];
        wave_min_multi<kWaves>(keys, m, L);
#pragma unroll
        for (int q = 0; q < kWaves; ++q) {
            const int a = lane_value(srow[q], L[q]), b = lane_value(snode[q], L[q]);
            if (lane == 0) { s_dec.sv[q] = m[q]; s_dec.srow[q] = a; s_dec.snode[q] = b; }
        }
    }
    AHC_STAMP(6);
    lds_barrier();
    const Dec dv = s_dec;  // one batch of LDS reads, everything below is register arithmetic on uniform values (moving it to the scalar
                           // unit with readfirstlane was measured 9 % slower: the chain is latency-bound on either unit)
    AHC_STAMP(7);
    // (a) finish the rows produced by the previous round
    double pd1[kPend];
    int pnn[kPend], pnnnode[kPend];
#pragma unroll
    for (int k = 0; k < kPend; ++k) {
        pd1[k] = dv.pd[k]; pnn[k] = dv.ps[k]; pnnnode[k] = dv.pn[k];
        if (!(pd1[k] < dinf())) { pnn[k] = -1; pnnnode[k] = -1; }
        if (st.pend_row[k] < 0) { pd1[k] = dinf(); pnn[k] = -1; pnnnode[k] = -1; }
        else {
#pragma unroll
            for (int j = 0; j < CPT; ++j)
                if (x0 + j == st.pend_row[k]) { rs[j].d1 = pd1[k]; rs[j].nn = pnn[k]; rs[j].nnnode = pnnnode[k]; e2x[j] = pd1[k]; }   // a scan yields no second minimum: the others are >= d1
        }
    }
    // (b) smallest row minimum (with its row) over all blocks and the finished rows; rows within 2 eps of it
    double g1 = dinf();
    int R1 = -1, Q1 = -1, NR1 = -1, NQ1 = -1;
    if (dv.v1 < g1) { g1 = dv.v1; R1 = dv.r1; Q1 = dv.q1; NR1 = dv.nr1; NQ1 = dv.nq1; }
#pragma unroll
    for (int k = 0; k < kPend; ++k) {
        const int P = st.pend_row[k];
        if (P >= 0 && lt2(pd1[k], P, g1, R1 < 0 ? INT_MAX : R1)) { g1 = pd1[k]; R1 = P; Q1 = pnn[k]; NR1 = st.pend_node[k]; NQ1 = pnnnode[k]; }
    }
    if (!(g1 < dinf())) R1 = -1;
    // The operands of the merge of (R1, Q1) are requested HERE: in all but a handful of rounds that pair is what the round merges, and the
    // window count, the state tests and the dispatch below (~900 cycles of branches on uniform values) only decide whether the values are
    // used.  The second memory round trip of the round starts that much earlier; a round that does something else drops them.
    constexpr int kCk = 4;                                   // centroid elements per lane handled without a loop (d <= 256)
    const bool spec = FA_AHC_SPECULATE && R1 >= 0 && Q1 >= 0;
    const bool spec_lo = R1 < Q1;
    const int sp_a = spec_lo ? R1 : Q1, sp_b = spec_lo ? Q1 : R1, sp_na = spec_lo ? NR1 : NQ1, sp_nb = spec_lo ? NQ1 : NR1;
    double sp_ma = 0.0, sp_mb = 0.0, sp_da[CPT], sp_db[CPT], sp_xa[kCk], sp_xb[kCk];
#pragma unroll
    for (int j = 0; j < kCk; ++j) { sp_xa[j] = 0.0; sp_xb[j] = 0.0; }
#pragma unroll
    for (int j = 0; j < CPT; ++j) { sp_da[j] = 0.0; sp_db[j] = 0.0; }
    // (the youngest load of the round's first batch is consumed here: the load counter completes in order, so everything older has arrived and
    // nothing in the decision below has to wait on the counter — a wait there would also wait for the requests that follow)
    asm volatile("" :: "v"(nanflag), "v"(e2x[CPT - 1]), "v"(rs[CPT - 1].d1), "v"(nx[CPT - 1]));
    if (spec) {
        // sizes and centroids first, the two matrix entries (a cold row each) last: loads complete in order.  (Requesting the entries from
        // every thread, so that the wait for the centroids need not cover them, was measured: dead columns then read cold column copies
        // nobody needs — 5.8 instead of 5.3 us per round at 43 200 points.)
        sp_ma = w.sizes[sp_na]; sp_mb = w.sizes[sp_nb];
        const double *ca = w.C + static_cast<size_t>(sp_na) * d, *cb = w.C + static_cast<size_t>(sp_nb) * d;
#pragma unroll
        for (int j = 0; j < kCk; ++j) { const int k = lane + 64 * j; sp_xa[j] = k < d ? ca[k] : 0.0; sp_xb[j] = k < d ? cb[k] : 0.0; }
        __builtin_amdgcn_sched_barrier(0);
        if (st.mode == FA_AHC_MODE_AUTO) {
            pair_entries<CPT>(w.M, Np, sp_a, sp_na, x0, nx, st.sym_limit, sp_a, sp_b, sp_da);
            pair_entries<CPT>(w.M, Np, sp_b, sp_nb, x0, nx, st.sym_limit, sp_a, sp_b, sp_db);
        }
    }
    const double glim = g1 + 2.0 * st.eps;
    int nwin = 0;  // conservative (never too small): nested counts were taken against local minima
    if (dv.v1 <= glim) nwin += dv.cnt;
#pragma unroll
    for (int k = 0; k < kPend; ++k) if (st.pend_row[k] >= 0 && pd1[k] <= glim) nwin += 1;

    AHC_STAMP(8);
    if (st.done || st.halt) {  // finished or waiting for the host: carry the state forward
        if (blk == 0 && tid == 0) {
            AhcHot n = st;
            n.prev_op = OP_NONE;
            for (int k = 0; k < kPend; ++k) n.pend_row[k] = -1;
            *nhot = n;
        }
        return;
    }
    Decision D;
    D.op = OP_NONE; D.a = D.b = D.na = D.nb = -1; D.dab = -1.0; D.lim = st.lim; D.halt = D.need_exact = D.error = D.done = 0;
    const WinCounters *cr = w.cnt + ((ph + 3) & 3);
    // The common case as ONE test (every round of a run but a handful): a merge of the certified pair (R1, Q1).  The general chain below
    // costs eight dependent compare-and-branch steps on uniform values before the operands of the merge can be requested.
    const bool plain_merge = !nanflag && st.step < N - 1 && st.prev_op != OP_COLLECT && st.prev_op != OP_PAIRS && R1 >= 0 && Q1 >= 0 &&
                             (st.mode == FA_AHC_MODE_EXACT || nwin == 2);
    if (plain_merge) {
        D.op = OP_MERGE;
    } else if (nanflag) {
        D.halt = 1; D.error = 1;  // NaN distance in an earlier round
    } else if (st.step >= N - 1) {
        D.done = 1;
    } else if (st.prev_op == OP_COLLECT) {
        const unsigned long long sk = cr->stale_key;
        const int nc = cr->ncand;
        if (sk != ~0ULL) { D.op = OP_RESCAN; D.a = static_cast<int>(sk >> 32); D.na = static_cast<int>(sk & 0xffffffffULL); }
        else if (nc > kMaxCand || nc < 1) { D.halt = 1; D.need_exact = 1; }
        else D.op = OP_PAIRS;
    } else if (st.prev_op == OP_PAIRS) {
        const int np = cr->npairs;
        if (np > kMaxPairs || np < 1) { D.halt = 1; D.need_exact = 1; }
        else {
            double best; int bp; bool tie;
            exact_min_pair(w, np, s_sq, s_val, s_idx, best, bp, tie);
            if (bp < 0) { D.halt = 1; D.error = 1; }               // a NaN distance among the window's pairs (nan_error of the reference)
            else if (bp == INT_MAX) { D.halt = 1; D.error = 3; }   // no pair at all: an internal selection failure, reported as such
            else if (tie) { D.halt = 1; D.need_exact = 2; }   // an EXACT tie at the minimum: which pair the reference takes is its heap's business -> reference order
            else { const int4 e = w.pairs[bp]; D.op = OP_MERGE; D.a = e.x; D.b = e.y; D.na = e.z; D.nb = e.w; D.dab = best; }
        }
    } else if (R1 < 0) {
        D.halt = 1; D.error = 2;  // cannot happen with finite data; stop rather than spin
    } else if (Q1 < 0) {
        D.op = OP_RESCAN; D.a = R1; D.na = NR1;  // a lower bound reached the minimum: re-scan that row first
    } else {
        // The pair (R1, Q1) is stored once, so row Q1 carries the same value: exactly two row minima inside the
        // window [g1, g1 + 2 eps] means {R1, Q1} is the unique candidate pair (any other entry <= lim of either row
        // would put a third row inside the window; bounds of stale rows count as row minima).  nwin == 2 (and exact rows) took the
        // branch at the top; here the window holds more: collect it.
        D.op = OP_COLLECT; D.lim = glim;
    }
    if (D.op == OP_MERGE && D.a < 0) {
        const bool lo = R1 < Q1;
        D.a = lo ? R1 : Q1; D.b = lo ? Q1 : R1; D.na = lo ? NR1 : NQ1; D.nb = lo ? NQ1 : NR1;
    }
    // rows produced this round: [0] the merged row / the forced re-scan, [1..] piggy-backed re-scans of the stale rows
    // with the smallest bounds (one candidate per wave's share of the blocks; a heuristic, any choice is correct)
    int prow[kPend], pnode_[kPend];
#pragma unroll
    for (int k = 0; k < kPend; ++k) { prow[k] = -1; pnode_[k] = -1; }
    if (D.op == OP_MERGE || D.op == OP_RESCAN) {
        prow[0] = D.a; pnode_[0] = D.op == OP_MERGE ? N + st.step : D.na;
        bool used[kWaves];
#pragma unroll
        for (int wv = 0; wv < kWaves; ++wv) {
            const int sr = dv.srow[wv];
            used[wv] = !(dv.sv[wv] < dinf()) || sr < 0 || sr == D.a || (D.op == OP_MERGE && sr == D.b);
        }
#pragma unroll
        for (int k = 1; k < kPend; ++k) {
            int bw = -1;
            double bv = dinf();
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) if (!used[wv] && dv.sv[wv] < bv) { bv = dv.sv[wv]; bw = wv; }
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) if (wv == bw) { used[wv] = true; prow[k] = dv.srow[wv]; pnode_[k] = dv.snode[wv]; }
        }
    }
    AHC_STAMP(1);

    // ---- phase 2 ------------------------------------------------------------------------------------------------
    bool dirty[CPT], e2_dirty[CPT], in_flight[CPT];   // in_flight: the row is being (re)produced: it leaves the record until the next round finishes it
    bool was_pending = false;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        bool wp = false;
#pragma unroll
        for (int k = 0; k < kPend; ++k) wp = wp || x0 + j == st.pend_row[k];
        dirty[j] = wp; e2_dirty[j] = wp; in_flight[j] = false;
        was_pending = was_pending || wp;
    }
    if (D.done || D.halt) {
        if (blk == 0 && tid == 0) {
            AhcHot n = st;
            n.done = D.done; n.halt = D.halt; n.need_exact = D.need_exact; n.error = D.error;
            n.prev_op = OP_NONE;
            for (int k = 0; k < kPend; ++k) n.pend_row[k] = -1;
            *nhot = n;
        }
        if (was_pending) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) if (dirty[j]) { w.row[x0 + j] = rs[j]; w.e2[x0 + j] = e2x[j]; }
        }
        return;
    }

    double pkey[kPend];  // the smallest of this thread's entries of each row being produced (and the slot / node it belongs to)
    int pslot[kPend], pnd[kPend];
#pragma unroll
    for (int k = 0; k < kPend; ++k) { pkey[k] = dinf(); pslot[k] = x0; pnd[k] = nx[0]; }

    if (D.op == OP_MERGE) {
        const int a = D.a, b = D.b, na = D.na, nb = D.nb, nnew = N + st.step;
        const bool sp_hit = spec && a == sp_a && b == sp_b && na == sp_na && nb == sp_nb;   // uniform; false only for the pair an exact window picked
        const double *ca = w.C + static_cast<size_t>(na) * d, *cb = w.C + static_cast<size_t>(nb) * d;
        bool act[CPT], any_act = false, all_act = true;
#pragma unroll
        for (int j = 0; j < CPT; ++j) { act[j] = nx[j] != kDead && x0 + j != a && x0 + j != b; any_act = any_act || act[j]; all_act = all_act && act[j]; }
        double ma = sp_ma, mb = sp_mb, da[CPT], db[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) { da[j] = sp_da[j]; db[j] = sp_db[j]; }
        if (!sp_hit) {
            ma = w.sizes[na]; mb = w.sizes[nb];
#pragma unroll
            for (int j = 0; j < CPT; ++j) { da[j] = 0.0; db[j] = 0.0; }
            if (st.mode == FA_AHC_MODE_AUTO) {  // valid copy of a pair lives in the row of the younger node
                pair_entries<CPT>(w.M, Np, a, na, x0, nx, st.sym_limit, a, b, da);
                pair_entries<CPT>(w.M, Np, b, nb, x0, nx, st.sym_limit, a, b, db);
            }
        }
        const double den = ma + mb;
        if constexpr (kPend > 1) {
#pragma unroll
            for (int k = 1; k < kPend; ++k) {  // piggy-backed re-scans: pairs not touched by this merge (CPT == 1 only)
                const int S = prow[k];
                if (S >= 0 && act[0] && x0 != S)
                    pkey[k] = pair_entry(w.M, Np, S, pnode_[k], x0, nx[0], st.sym_limit);
            }
        }
        // merged centroid (FastClusterWrapper.cpp:89-100), and |ca - cb|^2 summed as a tree (error <= ~10 ulp, independent of the merge
        // depth).  Every wave evaluates the whole sum: no workgroup barrier.  The centroid elements are REQUESTED together (an un-unrolled
        // loop made four dependent round trips of it: 3 800 of a round's 15 600 cycles), and the division runs only in the wave that
        // stores the centroid.
        double xa[kCk], xb[kCk];
#pragma unroll
        for (int j = 0; j < kCk; ++j) { xa[j] = sp_xa[j]; xb[j] = sp_xb[j]; }
        if (!sp_hit) {
#pragma unroll
            for (int j = 0; j < kCk; ++j) { const int k = lane + 64 * j; xa[j] = k < d ? ca[k] : 0.0; xb[j] = k < d ? cb[k] : 0.0; }
        }
        AHC_STAMP(9);
        const bool keeps_centroid = wave == 0 && (st.mode == FA_AHC_MODE_EXACT || blk == 0);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < kCk; ++j) {
            const int k = lane + 64 * j;
            if (keeps_centroid && k < d) {
                const double cc = __ddiv_rn(__dadd_rn(__dmul_rn(xa[j], ma), __dmul_rn(xb[j], mb)), den);
                if (st.mode == FA_AHC_MODE_EXACT) s_cvec[k] = cc;
                if (blk == 0) w.C[static_cast<size_t>(nnew) * d + k] = cc;
            }
            const double diff = xa[j] - xb[j];
            part += diff * diff;
        }
        for (int k = lane + 64 * kCk; k < d; k += 64) {      // d > 256
            const double ya = ca[k], yb = cb[k];
            if (keeps_centroid) {
                const double cc = __ddiv_rn(__dadd_rn(__dmul_rn(ya, ma), __dmul_rn(yb, mb)), den);
                if (st.mode == FA_AHC_MODE_EXACT) s_cvec[k] = cc;
                if (blk == 0) w.C[static_cast<size_t>(nnew) * d + k] = cc;
            }
            const double diff = ya - yb;
            part += diff * diff;
        }
        double dab = wave_sum(part);
        if (D.dab >= 0.0) dab = D.dab;
        AHC_STAMP(2);
        double dc[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) dc[j] = dinf();
        if (st.mode == FA_AHC_MODE_AUTO) {
            // Lance-Williams centroid update: a filter only, ties/near-ties are re-evaluated exactly.
            // weights from ONE division (the values are a filter, certified by the 2 eps window: the two extra roundings stay inside the
            // 16 u per merge level that eps budgets for 8); three IEEE fp64 divisions were ~40 dependent instructions per round
            const double inv = 1.0 / den, wa = ma * inv, wb = mb * inv, wab = wa * wb;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                if (act[j]) {
                    dc[j] = wa * da[j] + wb * db[j] - wab * dab;
                    if (!(dc[j] > 0.0)) dc[j] = 0.0;  // also keeps -0.0 out of the bit-pattern reductions
                }
            }
        } else {
            __syncthreads();
            double sum[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) sum[j] = 0.0;
#pragma unroll 8
            for (int k = 0; k < d; ++k) {
                double col[CPT];
                load_f64<CPT>(w.XT + static_cast<size_t>(k) * Np + x0, col);
                const double ck = s_cvec[k];
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    const double diff = __dsub_rn(ck, col[j]);
                    sum[j] = __dadd_rn(sum[j], __dmul_rn(diff, diff));  // sqeuclidean_extended (FastClusterWrapper.cpp:68-75): sequential in k per column
                }
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) if (act[j]) { dc[j] = sum[j]; if (sum[j] != sum[j]) w.flags[0] = 1; }
            __syncthreads();
            if (a / kCols == blk)
                for (int k = tid; k < d; k += kBlk) w.XT[static_cast<size_t>(k) * Np + a] = s_cvec[k];
        }
        if constexpr (CPT == 1) {
            const int x = x0;
            if (act[0]) {
                w.M[static_cast<size_t>(a) * Np + x] = dc[0];
                // Row x against its entry for the new cluster.  e2x bounds the entries of the row OTHER than the nearest neighbour's from below
                // (exact second minimum after the start-up scan, then maintained: an entry that appears lowers it, entries that disappear
                // leave it a bound).  It decides the case that used to make half of all rows stale on chaining data — the nearest neighbour
                // WAS one of the merged slots (every point's nearest neighbour is the growing cluster) and the new entry is larger than the
                // old minimum: if it is still below everything else (dc < e2x) the row simply keeps the cluster as its neighbour.
                const bool vld = rs[0].nn >= 0;
                const bool hit = vld && (rs[0].nn == a || rs[0].nn == b);
                if (!hit) {
                    if (dc[0] < rs[0].d1 || (vld && dc[0] == rs[0].d1 && a <= rs[0].nn)) {   // new minimum (a stale row: dc below its bound IS its minimum)
                        e2x[0] = rs[0].d1; rs[0].d1 = dc[0]; rs[0].nn = a; rs[0].nnnode = nnew; dirty[0] = true; e2_dirty[0] = true;
                    } else if (dc[0] < e2x[0]) { e2x[0] = dc[0]; e2_dirty[0] = true; }
                } else if (dc[0] < e2x[0]) {                                   // unique minimum again (strict: a tie goes to a re-scan)
                    rs[0].d1 = dc[0]; rs[0].nn = a; rs[0].nnnode = nnew; dirty[0] = true;
                } else {                                                        // minimum lost: every entry is >= min(e2x, dc) = e2x, a lower bound
                    rs[0].d1 = e2x[0]; rs[0].nn = -1; dirty[0] = true;
                }
                pkey[0] = dc[0];
#pragma unroll
                for (int k = 1; k < kPend; ++k)
                    if (x == prow[k]) { pkey[k] = dc[0]; pslot[k] = a; pnd[k] = nnew; in_flight[0] = true; }  // its entry for the new cluster
            } else if (x == a) {
                nx[0] = nnew; rs[0].d1 = dinf(); rs[0].nn = -1; rs[0].nnnode = -1; e2x[0] = dinf(); dirty[0] = true; e2_dirty[0] = true; in_flight[0] = true;
                w.sizes[nnew] = den;
                w.node[x] = nnew;
            } else if (x == b) {
                nx[0] = kDead; rs[0].d1 = dinf(); rs[0].nn = -1; rs[0].nnnode = -1; dirty[0] = true;
                w.node[x] = kDead;
            }
        } else {
            // Several slots per thread: the same rules as selects, no branch per column (the CPT chains interleave).  The new row leaves as ONE store
            // per thread: the entry of a dead or merged column is never read again (readers ask for live columns only; slot b stays dead, (a, a)
            // is no pair), so it is written as 0 rather than skipped.
            double dcs[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) dcs[j] = act[j] ? dc[j] : 0.0;
            store_f64<CPT>(w.M + static_cast<size_t>(a) * Np + x0, dcs);
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const bool vld = rs[j].nn >= 0;
                const bool hit = vld && (rs[j].nn == a || rs[j].nn == b);
                const bool below_e2 = dc[j] < e2x[j];
                const bool c_new = act[j] && !hit && (dc[j] < rs[j].d1 || (vld && dc[j] == rs[j].d1 && a <= rs[j].nn));   // new minimum
                const bool c_e2 = act[j] && !hit && !c_new && below_e2;                                                   // new second minimum
                const bool c_keep = act[j] && hit && below_e2;                                                            // unique minimum again
                const bool c_lost = act[j] && hit && !below_e2;                                                           // minimum lost: e2x is a lower bound
                const bool take = c_new || c_keep;
                const double d1_old = rs[j].d1, e2_old = e2x[j];
                e2x[j] = c_new ? d1_old : (c_e2 ? dc[j] : e2_old);
                rs[j].d1 = take ? dc[j] : (c_lost ? e2_old : d1_old);
                rs[j].nn = take ? a : (c_lost ? -1 : rs[j].nn);
                rs[j].nnnode = take ? nnew : rs[j].nnnode;
                dirty[j] = dirty[j] || take || c_lost;
                e2_dirty[j] = e2_dirty[j] || c_new || c_e2;
                const bool lower = act[j] && dc[j] < pkey[0];          // ascending j: the lowest slot keeps a tie
                pkey[0] = lower ? dc[j] : pkey[0]; pslot[0] = lower ? x0 + j : pslot[0]; pnd[0] = lower ? nx[j] : pnd[0];
            }
            if (a >= x0 && a < x0 + CPT) {          // the thread that owns slot a (one in the grid) — and the one that owns b
#pragma unroll
                for (int j = 0; j < CPT; ++j)
                    if (x0 + j == a) { nx[j] = nnew; rs[j].d1 = dinf(); rs[j].nn = -1; rs[j].nnnode = -1; e2x[j] = dinf(); dirty[j] = true; e2_dirty[j] = true; in_flight[j] = true; }
                w.sizes[nnew] = den;
                w.node[a] = nnew;
            }
            if (b >= x0 && b < x0 + CPT) {
#pragma unroll
                for (int j = 0; j < CPT; ++j)
                    if (x0 + j == b) { nx[j] = kDead; rs[j].d1 = dinf(); rs[j].nn = -1; rs[j].nnnode = -1; dirty[j] = true; }
                w.node[b] = kDead;
            }
        }
        if (blk == 0 && tid == 0) {
            double *z = w.Z + static_cast<size_t>(st.step) * 4;
            z[0] = na < nb ? na : nb;  // LinkageOutput::append (FastClusterWrapper.cpp:150-160)
            z[1] = na < nb ? nb : na;
            z[2] = 0.0;                // exact height filled by ahc_heights after the loop
            z[3] = den;
        }
    } else if (D.op == OP_RESCAN) {
#pragma unroll
        for (int k = 0; k < kPend; ++k) {
            const int S = prow[k];
            if (S < 0) continue;
            double ent[CPT];
            pair_entries<CPT>(w.M, Np, S, pnode_[k], x0, nx, st.sym_limit, S, -1, ent);
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                if (nx[j] != kDead && x0 + j != S && (CPT == 1 || ent[j] < pkey[k])) { pkey[k] = ent[j]; pslot[k] = x0 + j; pnd[k] = nx[j]; }
                if (x0 + j == S) in_flight[j] = true;
            }
        }
        // consumed inside the branch: a load still pending where the branches join makes the compiler wait on the in-order memory counter at
        // the join's first use of pkey — and on the MERGE path that wait finds only this round's STORES outstanding: a store round trip in
        // front of the block record of every merge round
#pragma unroll
        for (int k = 0; k < kPend; ++k) asm volatile("" :: "v"(pkey[k]));
    } else if (D.op == OP_COLLECT) {
        WinCounters *cw = w.cnt + (ph & 3);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            if (nx[j] != kDead && rs[j].d1 <= D.lim) {
                if (rs[j].nn < 0) atomicMin(&cw->stale_key, (static_cast<unsigned long long>(x0 + j) << 32) | static_cast<unsigned>(nx[j]));
                else { const int i = atomicAdd(&cw->ncand, 1); if (i < kMaxCand) w.cand[i] = make_int2(x0 + j, nx[j]); }
            }
        }
    } else if (D.op == OP_PAIRS) {
        WinCounters *cw = w.cnt + (ph & 3);
        const int nc = cr->ncand;
        for (int q = 0; q < nc; ++q) {
            const int2 cj = w.cand[q];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int x = x0 + j;
                if (nx[j] == kDead || x == cj.x) continue;
                const double val = pair_entry(w.M, Np, cj.x, cj.y, x, nx[j], st.sym_limit);
                if (val <= D.lim) {
                    const int slot = atomicAdd(&cw->npairs, 1);
                    if (slot < kMaxPairs) w.pairs[slot] = cj.x < x ? make_int4(cj.x, x, cj.y, nx[j]) : make_int4(x, cj.x, nx[j], cj.y);
                }
            }
        }
    }
    AHC_STAMP(3);

    // own row state back to HBM (only when it changed), then the record of the next round
    {
        bool any_e2 = false;
#pragma unroll
        for (int j = 0; j < CPT; ++j) any_e2 = any_e2 || e2_dirty[j];
        if (any_e2) store_f64<CPT>(w.e2 + x0, e2x);       // the thread's CPT bounds as one store (the unchanged ones rewrite their own value)
        if constexpr (CPT == 1) {
            if (dirty[0]) w.row[x0] = rs[0];
        } else {
            bool any_row = false;
#pragma unroll
            for (int j = 0; j < CPT; ++j) any_row = any_row || dirty[j];
            if (any_row) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) w.row[x0 + j] = rs[j];
            }
        }
    }
    double keys_all[CPT], key = dinf(), skey = dinf();
    int bx = x0, bnx = nx[0], bnn = rs[0].nn, bnnnode = rs[0].nnnode;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const bool live = nx[j] != kDead && !in_flight[j];
        keys_all[j] = live ? rs[j].d1 : dinf();
        if (j == 0 || keys_all[j] < key) { key = keys_all[j]; bx = x0 + j; bnx = nx[j]; bnn = rs[j].nn; bnnnode = rs[j].nnnode; }
        if (live && rs[j].nn < 0 && rs[j].d1 < skey) skey = rs[j].d1;
    }
    block_record<CPT>(w, npar, blk, st.eps, key, keys_all, skey, pkey, pslot, pnd, bx, bnx, bnn, bnnnode, s_out);
    AHC_STAMP(4);
    if (blk == 0 && tid == 0) {  // clear the window counters of the next round (here, in the tail: in front of the decision the store's round
        WinCounters *z = w.cnt + ((ph + 1) & 3);   // trip sat on the critical path of workgroup 0 — the next wait for a load also waits for it)
        z->stale_key = ~0ULL; z->ncand = 0; z->npairs = 0;
    }
    if (blk == 0 && tid == 0) {
        AhcHot n = st;
        n.prev_op = D.op;
        for (int k = 0; k < kPend; ++k) { n.pend_row[k] = prow[k]; n.pend_node[k] = pnode_[k]; if (k > 0 && prow[k] >= 0) bump(&w.state[0].piggy); }
        n.lim = D.lim;
        if (D.op == OP_MERGE) n.step = st.step + 1;
        n.rounds32 = st.rounds32 + 1;
        *nhot = n;
        if (D.op == OP_RESCAN) bump(&w.state[0].rescans);
        if (D.op == OP_COLLECT) bump(&w.state[0].windows);
    }
    AHC_STAMP(5);
#ifdef FA_AHC_PROFILE
    if (blk == prof_blk && tid == 0) {
        for (int i = 0; i < 12; ++i) atomicAdd(&w.prof[i], t_seg[i]);
        atomicAdd(&w.prof[15], 1ULL);
    }
#endif
}

// Entry kernels of the round.
//   ahc_round_t<false>: one problem, workspace in the kernel arguments.
//   ahc_round_t<true> : many problems; workgroup b looks up (problem, block) in a map and the problem's workspace in a table, both in HBM
//                       (constant address space = scalar loads): two dependent memory round trips before the round can start.
//   ahc_round_args    : up to kArgProblems problems with their workspaces and block ranges IN the kernel arguments: no extra round trip
//                       (fa_ahc_linkage_batch / fa_offline_cluster_batch with <= 16 recordings).
template <bool BATCH, bool BIG, int CPT = 1>
__global__ __launch_bounds__(kBlk) void ahc_round_t(const int ph, const int nblk_, AhcState *const state_, RecA *const recA_, int4 *const recI_, RecP *const recP_,
                                                    const unsigned off_row, const unsigned off_node, const unsigned off_e2, const unsigned off_flags,
                                                    const Ws w_one, const Ws *__restrict__ table, const int2 *__restrict__ blkmap) {
    // The leading scalar arguments repeat what the FIRST loads of a round need (round parity, block count, state and record arrays, and the
    // row-state arrays as byte offsets from the state): scalars at the front of the argument list are PRELOADED into SGPRs with the
    // wavefront (Makefile: -amdgpu-kernarg-preload-count; a by-value struct is not), so nothing of the round's first memory round trip
    // waits for a scalar load of the argument segment (the compiler had put that wait in front of the record requests).
    int blk_ = blockIdx.x;
    Ws w_ = w_one;
    if (!BATCH) {
        w_.nblk = nblk_; w_.Np = nblk_ * kBlk * CPT; w_.state = state_; w_.recA = recA_; w_.recI = recI_; w_.recP = recP_;
        char *base = reinterpret_cast<char *>(state_);
        w_.row = reinterpret_cast<RowSt *>(base + off_row); w_.node = reinterpret_cast<int *>(base + off_node);
        w_.e2 = reinterpret_cast<double *>(base + off_e2); w_.flags = reinterpret_cast<int *>(base + off_flags);
    }
    if (BATCH) {
        static_assert(sizeof(Ws) % 8 == 0, "Ws is copied as 64-bit words");
        typedef const int __attribute__((address_space(4))) *c_i32;
        typedef const unsigned long long __attribute__((address_space(4))) *c_u64;
        const int prob = ((c_i32)reinterpret_cast<const int *>(blkmap))[2 * blockIdx.x];
        blk_ = ((c_i32)reinterpret_cast<const int *>(blkmap))[2 * blockIdx.x + 1];
        unsigned long long words[sizeof(Ws) / 8];
        c_u64 src = (c_u64)reinterpret_cast<const unsigned long long *>(table) + static_cast<size_t>(prob) * (sizeof(Ws) / 8);
#pragma unroll
        for (unsigned i = 0; i < sizeof(Ws) / 8; ++i) words[i] = src[i];
        __builtin_memcpy(&w_, words, sizeof(Ws));
    }
    ahc_round_body<false, BIG, CPT>(w_, blk_, ph);
}

// A problem of at most 256 points is ONE block: its rounds need no device-wide barrier at all, a workgroup barrier between them (with
// the release / acquire that makes the records and row states written by some threads visible to the others) is enough — all rounds
// of a replay in one launch, no kernel boundary, operands in the local caches (agent-scope fences around the barrier were measured
// 0.5 us per round slower and are not needed inside one workgroup).
template <int CPT>   // up to kBlk * CPT points
__global__ __launch_bounds__(kBlk) void ahc_rounds_single_block(const Ws w, const int rounds) {
    for (int r = 0; r < rounds; ++r) {
        ahc_round_body<false, false, CPT>(w, 0, r & 3);
        __syncthreads();   // workgroup-scope release / acquire: the waves of one workgroup share the CU's caches
    }
}

// ahc_round_uni: K problems in ONE launch without any look-up in front of the round (round 4).  The host lays the K workspaces out with the
// SAME layout (that of the largest problem; a smaller one simply has more dead padding slots) at a constant stride, so every array of
// problem k is the array of problem 0 + k * stride: the grid is (blocks, problems), the problem index is the workgroup id in y (an SGPR the
// hardware hands over), and the addresses of the round's first memory round trip are arithmetic on PRELOADED kernel arguments — the same
// zero scalar round trips as the single-problem kernel.  (ahc_round_args, the round-2 form: 154 scalar instructions and three dependent
// scalar-cache round trips — block -> problem search over 16 block ranges, then two batches of workspace fields out of a by-value array
// indexed by the problem — in front of its first request: 11 us per round of 16 problems against 5.3 us for one.)  Only N differs per
// problem: it comes from the hot part of the problem's state, with the first batch of loads.
// arg 0 = (blocks << 2) | (round & 3), stride in 4 KB pages: 14 preloaded dwords like ahc_round_t.
template <int CPT>
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
    ahc_round_body<true, false, CPT>(w_, blockIdx.x, static_cast<int>(nblk_ph & 3u), sh);   // the host sends problems of more than 65 536 points elsewhere
}
#define FA_AHC_UNI_KERNEL(NAME, ATTR, CPT)                                                                                                              \
    __global__ __launch_bounds__(kBlk) ATTR void NAME(const unsigned nblk_ph, const unsigned stride_pages, AhcState *const state_, RecA *const recA_,  \
                                                      int4 *const recI_, RecP *const recP_, const unsigned off_row, const unsigned off_node,           \
                                                      const unsigned off_e2, const unsigned off_flags, const Ws w_one) {                              \
        ahc_round_uni_body<CPT>(nblk_ph, stride_pages, state_, recA_, recI_, recP_, off_row, off_node, off_e2, off_flags, w_one);                      \
    }
// one slot per thread at three register budgets (more co-resident workgroups per CU against spills; round 4) and the round-5 forms with 2 / 4 slots per
// thread (which one serves a batch: ahc_batch_uniform)
FA_AHC_UNI_KERNEL(ahc_round_uni, , 1)
FA_AHC_UNI_KERNEL(ahc_round_uni_w3, __attribute__((amdgpu_waves_per_eu(6, 6))), 1)   // "w3" / "w4": the second and third budget
FA_AHC_UNI_KERNEL(ahc_round_uni_w4, __attribute__((amdgpu_waves_per_eu(8, 8))), 1)
FA_AHC_UNI_KERNEL(ahc_round_uni_c2, , 2)
FA_AHC_UNI_KERNEL(ahc_round_uni_c4, , 4)

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

// ------------------------------------------------------------------------------ reference order (ahc_reforder.h)
// The run that reproduces the reference's choice among EXACTLY tied distances: every distance the reference evaluates is evaluated
// here (its sequential fp64 sums), in parallel over the active clusters, and ONE thread replays its selection (binary heap, active
// list, fa_ro::Sel).  Per dendrogram row: ro_scan (all workgroups: the new node against every active node, or the re-scan of a heap
// top whose neighbour is gone; block minima by (value, node id)) + ro_select (one workgroup: the minimum of the block minima, then
// the heap / list updates and the next pair).  O(N d) per merge and two dependent launches: ~40 us per merge instead of 6 — the price
// of the reference's order, paid only by inputs that contain exact ties at the minimum.
struct RoPart { double v; int32_t node, pad; };
struct RoDev {                       // scalars of fa_ro::Sel between launches + flags
    int32_t heap_size, list_first, merges, op, a, b, n, done, nan_seen, pad;
    long long scans;
};
struct RoWs {
    double *C, *XT, *sizes, *key, *pair_a, *pair_b, *height_sq, *Z;
    int32_t *node, *slot_of, *at, *pos, *nghbr, *next, *prev, *flags;
    RoPart *part;
    RoDev *dev;
    int32_t N, Np, d, nblk;
};

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

// ------------------------------------------------------------------------------ reference order, the matrix as the filter of its scans (round 5)
// The run above evaluates O(A d) exact sums per dendrogram row and lets ONE thread replay the reference's heap: 28-30 us per row at 43 200 x 256
// (profiles/r05_ties_probe.txt), what an input with exact ties paid for the reference's row order.  Here, whenever the N x N workspace is to be had:
//   * the scans ask the Lance-Williams matrix of the filter-based rounds (Gram-form start-up on the fp64 matrix cores, pair_entry's validity rule,
//     one row rewritten per merge): rom_scan writes the new row / reads the row of a re-scanned node — O(A) — and reduces it to block minima;
//   * rom_select, one wavefront: every entry within 2 eps of the smallest one is a candidate (eps bounds |entry - the reference's sum| as in
//     ahc_set_eps, plus the term of the sequentially summed d(a, b) used here); the candidates — one, on tie-free rows — are evaluated with the
//     reference's sequential sums (lane i sums candidate i; the squares are formed by the whole wavefront), the winner by (value, node id) is what the
//     reference's strict `<` scan in index order finds.  More than kRomCap candidates (massively duplicated inputs): the row is scanned again with
//     exact sums by every workgroup (kind ROM_EXACT — the scan of the run above);
//   * the heap replay is ahc_reforder.h's HeapK: entries carry their key, a sift works on a block fetched by the 64 lanes at once; every lane
//     executes every store of the selection (same address, same value), so whatever a lane reads later it has written itself.
// tests/cpu/ahc_rom_emul.cpp replays exactly this on the CPU against the reference build.
enum : int32_t { ROM_NEW = 0, ROM_RESCAN = 1, ROM_EXACT = 2 };
constexpr int kRomCap = 128;     // candidates one selection evaluates
constexpr int kRomBatch = 16;    // candidates summed side by side (one lane each)
constexpr int kRomChunk = 256;   // coordinates per staging pass
struct __attribute__((aligned(16))) RomPart { double v1; int32_t x1, n1; };   // smallest entry of the block by (value, node id): value, slot, node
struct RomDev {
    int32_t heap_size, list_first, merges, op, a, b, n, done;              // fa_ro::SelT between launches
    int32_t nan_seen, kind, scanned, sa, sb, created, pad0, pad1;          // what the next scan launch computes: the row of node `scanned` (slot sa)
    double ma, mb, dab, eps;                                               // ROM_NEW: sizes of a and b, their exact squared distance
    long long scans, exact_scans, cands, pad2;
};
static_assert(sizeof(RomDev) % 16 == 0, "copied in 16-byte pieces");
struct RomWs {
    double *M, *C, *XT, *sizes, *pair_a, *pair_b, *height_sq, *part2;
    fa_ro::Ent *ent;
    int32_t *node, *slot_of, *pos, *nghbr, *next, *prev, *flags;
    RomPart *part;
    RomDev *dev;
    unsigned long long *prof;   // [16] clock sums of the selection's phases (FA_ROM_PROFILE builds only)
    int32_t N, Np, d, nblk;
};

// what the scan after `sel` has to compute (host: the first one; device: every later one)
template <class S>
__host__ __device__ inline void rom_prepare(RomDev &st, const S &sel, const int32_t *slot_of, const double *sizes) {
    st.heap_size = sel.heap.size; st.list_first = sel.list.first; st.merges = sel.merges; st.op = sel.op; st.a = sel.a; st.b = sel.b; st.n = sel.n;
    if (sel.op == fa_ro::RO_NEW_ROW) {
        st.kind = ROM_NEW; st.created = sel.n + sel.merges - 1; st.scanned = st.created;
        st.sa = slot_of[sel.a]; st.sb = slot_of[sel.b]; st.ma = sizes[sel.a]; st.mb = sizes[sel.b]; st.dab = sel.height_sq[sel.merges - 1];
    } else if (sel.op == fa_ro::RO_RESCAN) {
        st.kind = ROM_RESCAN; st.scanned = sel.a; st.sa = slot_of[sel.a]; st.sb = -1; st.created = -1;
    } else st.done = 1;
}

__global__ __launch_bounds__(kBlk) void rom_scan(const RomWs w, const int ph) {
    extern __shared__ double s_c[];            // [d] coordinates of the scanned node (ROM_EXACT)
    __shared__ double s_v1[kWaves], s_v2[kWaves];
    __shared__ int s_x1[kWaves], s_n1[kWaves];
    const RomDev st = w.dev[ph];
    if (st.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x, x = blk * kBlk + tid, Np = w.Np, d = w.d;
    if (blk == w.nblk) {                       // one workgroup beyond the columns: the merged centroid (FastClusterWrapper.cpp:89-100), by node id and into
        if (st.kind != ROM_NEW) return;        // the slot-major transpose — its own two round trips, beside the row's instead of behind them in the block of slot sa
        const double *ca = w.C + static_cast<size_t>(st.a) * d, *cb = w.C + static_cast<size_t>(st.b) * d, den = st.ma + st.mb;
        for (int k = tid; k < d; k += kBlk) {
            const double cc = __ddiv_rn(__dadd_rn(__dmul_rn(ca[k], st.ma), __dmul_rn(cb[k], st.mb)), den);
            w.C[static_cast<size_t>(st.created) * d + k] = cc;
            w.XT[static_cast<size_t>(k) * Np + st.sa] = cc;
        }
        return;
    }
    double v = dinf();
    int nx;
    if (st.kind == ROM_NEW) {                  // Lance-Williams row of the node created from (a, b) into the row of slot sa
        // This run keeps the matrix SYMMETRIC over the live slots (the mirror workgroups of rom_select write the column of every new row while the
        // selection runs), so both entries are row copies: two coalesced requests next to node[x], nothing behind it.  (The filter-based rounds
        // read a column copy for every column younger than a — one 8-byte request per lane, each in another 345 KB row.)
        double *const ra = w.M + static_cast<size_t>(st.sa) * Np + x;
        const double da = *ra, db = w.M[static_cast<size_t>(st.sb) * Np + x];
        nx = w.node[x];
        if (nx != kDead && x != st.sa && x != st.sb) {
            const double den = st.ma + st.mb, inv = 1.0 / den, wa = st.ma * inv, wb = st.mb * inv, wab = wa * wb;
            v = wa * da + wb * db - wab * st.dab;
            if (!(v > 0.0)) v = 0.0;
            *ra = v;
        }
        if (x == st.sa) { w.node[x] = st.created; w.slot_of[st.created] = x; w.sizes[st.created] = st.ma + st.mb; }
        if (x == st.sb) w.node[x] = kDead;
    } else if (st.kind == ROM_RESCAN) {        // the row of a against every older node
        const double e = w.M[static_cast<size_t>(st.sa) * Np + x];
        nx = w.node[x];
        if (nx != kDead && x != st.sa && nx < st.scanned) v = e;
    } else {                                   // ROM_EXACT: the reference's sums of node `scanned` against every active node below it (ro_scan's)
        nx = w.node[x];
        const double *cs = w.C + static_cast<size_t>(st.scanned) * d;
        for (int k = tid; k < d; k += kBlk) s_c[k] = cs[k];
        __syncthreads();
        if (nx != kDead && x != st.sa && nx < st.scanned) {
            const double *col = w.XT + x;
            double sum = 0.0;
#pragma unroll 8
            for (int k = 0; k < d; ++k) {
                const double diff = __dsub_rn(col[static_cast<size_t>(k) * Np], s_c[k]);
                sum = __dadd_rn(sum, __dmul_rn(diff, diff));
            }
            if (sum != sum) w.flags[0] = 1;
            v = sum;
        }
    }
    // block minimum by (value, node id) + the block's second smallest value
    const double m = wave_min(v == v ? v : dinf());
    const bool fin = m < dinf();
    const unsigned id = wave_umin((fin && v == m) ? static_cast<unsigned>(nx) : static_cast<unsigned>(INT_MAX));
    const unsigned long long msk = __builtin_amdgcn_ballot_w64(fin && v == m && static_cast<unsigned>(nx) == id);
    const int L = __builtin_amdgcn_readfirstlane(msk ? __ffsll(static_cast<long long>(msk)) - 1 : 0);
    const int x1 = lane_value(x, L);
    const double second = wave_min((lane == L || v != v) ? dinf() : v);
    if (lane == 0) { s_v1[wave] = m; s_v2[wave] = second; s_x1[wave] = x1; s_n1[wave] = static_cast<int>(id); }
    lds_barrier();
    if (tid != 0) return;
    double bv = s_v1[0], b2 = s_v2[0];
    int bn = s_n1[0], bx = s_x1[0];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) {
        if (lt2(s_v1[wv], s_n1[wv], bv, bn)) { if (bv < b2) b2 = bv; bv = s_v1[wv]; bn = s_n1[wv]; bx = s_x1[wv]; if (s_v2[wv] < b2) b2 = s_v2[wv]; }
        else if (s_v1[wv] < b2) b2 = s_v1[wv];
    }
    RomPart pt; pt.v1 = bv; pt.x1 = bx; pt.n1 = bn;
    w.part[blk] = pt;
    w.part2[blk] = b2;
}

// Start-up of the matrix-filtered run: the reference's nearest LOWER-indexed neighbour of every point (fastcluster_internal.hpp:1653-1678) with the
// Gram-form matrix as the filter — row i: the smallest entry left of the diagonal, every entry within 2 eps of it is a candidate, the candidates (one, on
// tie-free rows) get the reference's sequential sum, lowest (value, index) wins.  Reads the lower triangle twice (2 x 7.5 GB at 43 200 points) where
// ro_lower_minima_direct evaluates all N^2 / 2 sums (57 ms there).  One workgroup per row.
__global__ __launch_bounds__(kBlk) void rom_lower_minima(const RomWs w, const AhcState *__restrict__ state, double *__restrict__ key) {
    __shared__ double s_v[kWaves];
    __shared__ int s_i[kWaves];
    const int i = blockIdx.x + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = w.d;
    const double *row = w.M + static_cast<size_t>(i) * w.Np;
    double mv = dinf();
    for (int j = tid; j < i; j += kBlk) { const double e = row[j]; if (e < mv) mv = e; }
    mv = wave_min(mv);
    if (lane == 0) s_v[wave] = mv;
    __syncthreads();
    double m = s_v[0];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) if (s_v[wv] < m) m = s_v[wv];
    __syncthreads();
    const double dmax = __longlong_as_double(static_cast<long long>(state[0].dmax_bits)), nmax = __longlong_as_double(static_cast<long long>(state[0].nmax_bits));
    const double u = 1.1102230246251565e-16;
    const double lim = m + 2.0 * (16.0 * static_cast<double>(w.N) * u * dmax + 8.0 * (static_cast<double>(d) + 2.0) * u * nmax);   // ahc_set_eps (a superset of the Gram term alone)
    const double *xi = w.C + static_cast<size_t>(i) * d;
    double best = dinf();
    int arg = INT_MAX;
    for (int j = tid; j < i; j += kBlk) {
        if (!(row[j] <= lim)) continue;
        const double *xj = w.C + static_cast<size_t>(j) * d;
        double sum = 0.0;
#pragma unroll 8
        for (int k = 0; k < d; ++k) { const double diff = __dsub_rn(xi[k], xj[k]); sum = __dadd_rn(sum, __dmul_rn(diff, diff)); }   // FastClusterWrapper.cpp:45-52
        if (sum != sum) w.flags[0] = 1;
        else if (lt2(sum, j, best, arg)) { best = sum; arg = j; }   // j ascending per thread
    }
    const double bm = wave_min(best);
    const unsigned bi = wave_umin((best == bm && bm < dinf()) ? static_cast<unsigned>(arg) : static_cast<unsigned>(INT_MAX));
    if (lane == 0) { s_v[wave] = bm; s_i[wave] = static_cast<int>(bi); }
    __syncthreads();
    if (tid != 0) return;
    double bv = s_v[0];
    int ba = s_i[0];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) if (lt2(s_v[wv], s_i[wv], bv, ba)) { bv = s_v[wv]; ba = s_i[wv]; }
    key[i] = bv;
    w.nghbr[i] = ba == INT_MAX ? 0 : ba;
}

struct WaveMem {   // ahc_reforder.h's block fetches by the 64 lanes of the selecting wavefront
    fa_ro::Ent *buf;   // LDS [fa_ro::kTreeEnts + 1]
    __device__ __forceinline__ static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void fetch_chain(const fa_ro::Ent *ent, const int32_t place, const int32_t depth) {
        const int lane = threadIdx.x & 63;
        wave_sync();
        if (lane < depth) buf[lane] = ent[fa_ro::heap_ancestor(place, lane)];
        wave_sync();
    }
    __device__ __forceinline__ void fetch_tree(const fa_ro::Ent *ent, const int32_t root, const int32_t size) {
        const int lane = threadIdx.x & 63;
        constexpr int kPer = (fa_ro::kTreeEnts - 1 + 63) / 64;   // entries per lane: all requested before the first one is stored (places beyond the heap ask for entry 0)
        typedef int v4i32 __attribute__((ext_vector_type(4)));    // an entry as one 16-byte register value (an array of the struct went through scratch memory)
        static_assert(sizeof(fa_ro::Ent) == sizeof(v4i32), "an entry is one 16-byte word");
        v4i32 v[kPer];
        wave_sync();
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int32_t t = 1 + lane + 64 * q, lev = 31 - __clz(t + 1);
            const int64_t p = ((static_cast<int64_t>(root) + 1) << lev) - 1 + (t + 1 - (1 << lev));   // fa_ro::heap_tree_place(root, t)
            v[q] = *reinterpret_cast<const v4i32 *>(ent + ((t < fa_ro::kTreeEnts && p < size) ? p : 0));
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q) { const int32_t t = 1 + lane + 64 * q; if (t < fa_ro::kTreeEnts) *reinterpret_cast<v4i32 *>(buf + t) = v[q]; }
        wave_sync();
    }
    __device__ __forceinline__ double key_at(const int32_t j) const { return buf[j].key; }
    __device__ __forceinline__ fa_ro::Ent ent_at(const int32_t j) const { return buf[j]; }
};

#ifdef FA_ROM_PROFILE   // where a selection spends its time: every stamp drains the memory counters first (phase costs in isolation)
#define ROM_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_now = clock64(); t_seg[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define ROM_STAMP(i) do {} while (0)
#endif

// keeps a value requested early alive up to here without using it (the request warmed the caches for the loads of the selection)
template <class T> __device__ __forceinline__ void rom_sink(const T v) { asm volatile("" ::"v"(v)); }

// Workgroup 0 (one wavefront) is the selection.  Workgroups 1 .. nblk mirror the row rom_scan has just written into its column, M[x][sa] = M[sa][x]:
// 8-byte stores into 43 200 different rows that nobody waits for — they drain while the selection walks its heap, and the next rom_scan finds every
// pair in BOTH orientations.  The state is double buffered by launch parity: the mirror workgroups read the record the selection does not write.
__global__ __launch_bounds__(64) void rom_select(const RomWs w, const int ph) {
    __shared__ fa_ro::Ent s_buf[fa_ro::kTreeEnts + 1];
    __shared__ __attribute__((aligned(16))) double s_t[kRomBatch][kRomChunk + 2];
    __shared__ int s_cand[kRomCap];
    RomDev st = w.dev[ph];
    if (st.done) return;
    const int lane = threadIdx.x, Np = w.Np, d = w.d, nblk = w.nblk;
    if (blockIdx.x > 0) {
        if (st.kind != ROM_NEW) return;
        const int x0 = (static_cast<int>(blockIdx.x) - 1) * kBlk + lane;
        const double *row = w.M + static_cast<size_t>(st.sa) * Np;
        double e[kBlk / 64];
        int nxs[kBlk / 64];
#pragma unroll
        for (int j = 0; j < kBlk / 64; ++j) { e[j] = row[x0 + 64 * j]; nxs[j] = w.node[x0 + 64 * j]; }
#pragma unroll
        for (int j = 0; j < kBlk / 64; ++j) { const int x = x0 + 64 * j; if (nxs[j] != kDead && x != st.sa) w.M[static_cast<size_t>(x) * Np + st.sa] = e[j]; }
        return;
    }
    const int flag0 = w.flags[0];              // requested with everything else; a NaN met by THIS launch is carried in `nan_here`
    bool nan_here = false;
#ifdef FA_ROM_PROFILE
    unsigned long long t_seg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = clock64();
    const unsigned long long w_begin = wall_clock64();
#endif
    // ---- requests whose addresses the state already holds, all in flight together: the block minima, the coordinates of the scanned node, and — values
    // not used here, the lines are what counts — the words the heap replay will ask for first: pos[] of the node it removes and of a, the last entry, the
    // top block of the heap
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    const int n2 = 2 * st.n - 1;               // node ids 0 .. 2 n - 2
    const int hs1 = st.heap_size > 1 ? st.heap_size - 1 : 0;   // last place of the heap
    const int rm_node = st.op == fa_ro::RO_NEW_ROW ? (st.b < st.list_first ? st.list_first : st.b) : st.a;   // the node heap.remove will be asked for (:1792-1797)
    const int warm_pos = w.pos[lane == 0 ? rm_node : st.a];    // lane 0: its place; issued FIRST, so it is here when the block minima are
    constexpr int kFast = 4;                   // block records per lane held in registers (N <= 65 536)
    const bool fast = nblk <= 64 * kFast;
    RomPart pr[kFast];
    double p2[kFast];
#pragma unroll
    for (int j = 0; j < kFast; ++j) {
        const int b = lane + 64 * j, bc = b < nblk ? b : nblk - 1;
        pr[j] = w.part[bc]; p2[j] = w.part2[bc];
        if (b >= nblk) { pr[j].v1 = dinf(); pr[j].n1 = INT_MAX; p2[j] = dinf(); }
    }
    const double *cs = w.C + static_cast<size_t>(st.scanned) * d;
    double xs0[kRomChunk / 64];
#pragma unroll
    for (int j = 0; j < kRomChunk / 64; ++j) { const int k = lane + 64 * j; xs0[j] = cs[k < d ? k : d - 1]; }
    constexpr int kWarmPer = (fa_ro::kTreeEnts - 1 + 63) / 64;
    v4i32 warm_top[kWarmPer];                  // places 1 .. 510: the block heap.replace walks first (entry 0 is replaced)
#pragma unroll
    for (int q = 0; q < kWarmPer; ++q) { const int pl = 1 + lane + 64 * q; warm_top[q] = *reinterpret_cast<const v4i32 *>(w.ent + (pl < hs1 ? pl : hs1)); }
    const v4i32 warm_last = *reinterpret_cast<const v4i32 *>(w.ent + hs1);

    ROM_STAMP(0);                              // the first round trip: block minima, coordinates, the warmed words
    // ---- second layer, in flight while the candidates are collected and their rows travel: around the place heap.remove starts from (the entry, its
    // ancestor chain, the block below it), and the list / slot words of the two nodes under the heap top — one of them, or the node created now, is the
    // next top (advance)
    const int rm_place = __builtin_amdgcn_readfirstlane(warm_pos);
    const int rp = rm_place >= 0 && rm_place <= hs1 ? rm_place : 0;
    const int rdepth = 31 - __clz(rp + 1);
    const v4i32 warm_chain = *reinterpret_cast<const v4i32 *>(w.ent + (lane < rdepth ? static_cast<int>((static_cast<unsigned>(rp) + 1u) >> (lane + 1)) - 1 : rp));
    v4i32 warm_tree[kWarmPer];
#pragma unroll
    for (int q = 0; q < kWarmPer; ++q) {
        const int32_t t = 1 + lane + 64 * q, lev = 31 - __clz(t + 1);
        const int64_t pl = ((static_cast<int64_t>(rp) + 1) << lev) - 1 + (t + 1 - (1 << lev));
        warm_tree[q] = *reinterpret_cast<const v4i32 *>(w.ent + ((t < fa_ro::kTreeEnts && pl < hs1) ? pl : 0));
    }
    int wn = warm_top[0].z;                    // lanes 0 / 1: the nodes at places 1 / 2 (the third word of an entry is its node)
    wn = (lane < 2 && lane + 1 < hs1 && wn >= 0 && wn < n2) ? wn : st.a;
    const int warm_ng = w.nghbr[wn], warm_nx = w.next[wn], warm_pv = w.prev[wn], warm_so = w.slot_of[wn];
    const double warm_sz = w.sizes[wn];
    fa_ro::SelT<fa_ro::HeapK<WaveMem>> sel;
    sel.heap.ent = w.ent; sel.heap.pos = w.pos; sel.heap.size = st.heap_size; sel.heap.mem.buf = s_buf;
    sel.list.next = w.next; sel.list.prev = w.prev; sel.list.first = st.list_first;
    sel.nghbr = w.nghbr; sel.n = st.n; sel.merges = st.merges; sel.op = st.op; sel.a = st.a; sel.b = st.b;
    sel.pair_a = w.pair_a; sel.pair_b = w.pair_b; sel.height_sq = w.height_sq;
    double best = dinf();
    int best_id = INT_MAX;
    if (st.kind == ROM_EXACT) {                // the block minima are the reference's sums: lowest (value, node id)
        sel.scan_begin();
        double v = dinf();
        int id = INT_MAX;
        for (int b = lane; b < nblk; b += 64) { const RomPart pt = w.part[b]; if (lt2(pt.v1, pt.n1, v, id)) { v = pt.v1; id = pt.n1; } }
        best = wave_min(v == v ? v : dinf());
        best_id = static_cast<int>(wave_umin((v == best && best < dinf()) ? static_cast<unsigned>(id) : static_cast<unsigned>(INT_MAX)));
    } else {
        double mv = dinf();
        if (fast) {
#pragma unroll
            for (int j = 0; j < kFast; ++j) if (pr[j].v1 < mv) mv = pr[j].v1;
        } else
            for (int b = lane; b < nblk; b += 64) { const double v1 = w.part[b].v1; if (v1 < mv) mv = v1; }
        const double m = wave_min(mv);
        int ncand = 0;
        bool collected = false;
        if (fast && m < dinf()) {              // the common row: ONE block minimum inside the window and that block's second entry outside it
            const double lim = m + 2.0 * st.eps;
            bool hit = false, dense = false;
            int n1 = INT_MAX;
#pragma unroll
            for (int j = 0; j < kFast; ++j) { const bool h = pr[j].v1 <= lim; if (h) { n1 = pr[j].n1; dense = dense || hit || p2[j] <= lim; hit = true; } }
            const unsigned long long mh = __builtin_amdgcn_ballot_w64(hit), mdn = __builtin_amdgcn_ballot_w64(dense);
            if (mdn == 0 && __popcll(mh) == 1) {
                if (hit) s_cand[0] = n1;
                ncand = 1;
                collected = true;
            }
        }
        if (!collected && m < dinf()) {
            const double lim = m + 2.0 * st.eps;
            const double *row = w.M + static_cast<size_t>(st.sa) * Np;
            for (int base = 0; base < nblk && ncand <= kRomCap; base += 64) {
                const int b = base + lane;
                RomPart pt; pt.v1 = dinf(); pt.x1 = -1; pt.n1 = INT_MAX;
                double v2 = dinf();
                if (fast) {
#pragma unroll
                    for (int j = 0; j < kFast; ++j) if (base == 64 * j) { pt = pr[j]; v2 = p2[j]; }
                } else if (b < nblk) { pt = w.part[b]; v2 = w.part2[b]; }
                const bool hit = pt.v1 <= lim, dense = hit && v2 <= lim, single = hit && !dense;
                const unsigned long long ms = __builtin_amdgcn_ballot_w64(single);
                const int at = ncand + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(ms >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(ms), 0));
                if (single && at < kRomCap) s_cand[at] = pt.n1;
                ncand += __popcll(ms);
                unsigned long long md = __builtin_amdgcn_ballot_w64(dense);
                while (md && ncand <= kRomCap) {   // a block with several entries inside the window: its 256 entries again
                    const int bb = base + __ffsll(static_cast<long long>(md)) - 1;
                    md &= md - 1;
                    int nxs[kBlk / 64];
                    double es[kBlk / 64];
#pragma unroll
                    for (int j = 0; j < kBlk / 64; ++j) { const int x = bb * kBlk + 64 * j + lane; nxs[j] = w.node[x]; es[j] = row[x]; }
#pragma unroll
                    for (int j = 0; j < kBlk / 64; ++j) {
                        const int x = bb * kBlk + 64 * j + lane, nx = nxs[j];
                        const bool c = nx != kDead && x != st.sa && nx < st.scanned && es[j] <= lim;
                        const unsigned long long mc = __builtin_amdgcn_ballot_w64(c);
                        const int ac = ncand + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mc >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mc), 0));
                        if (c && ac < kRomCap) s_cand[ac] = nx;
                        ncand += __popcll(mc);
                    }
                }
            }
        }
        if (ncand > kRomCap) {                 // too many for one wavefront: the same row by exact sums of every workgroup, then back here
            rom_sink(warm_pos); rom_sink(xs0[0]); rom_sink(warm_last.x); rom_sink(warm_chain.x); rom_sink(warm_top[0].x); rom_sink(warm_tree[0].x); rom_sink(warm_ng); rom_sink(warm_sz);
            if (lane == 0) { st.kind = ROM_EXACT; st.exact_scans = st.exact_scans + 1; w.dev[ph ^ 1] = st; }
            return;
        }
        st.cands = st.cands + ncand;
        WaveMem::wave_sync();
        ROM_STAMP(1);                          // candidates collected
        // third layer: the list / slot words of the recorded neighbours of those two nodes, of the first candidate (the neighbour the created node will
        // record, most rows) and of the created node itself
        int wo = lane < 2 ? warm_ng : (lane == 2 ? s_cand[0] : (st.created >= 0 ? st.created : st.a));
        wo = (wo >= 0 && wo < n2) ? wo : st.a;
        const int warm_onx = w.next[wo], warm_opv = w.prev[wo], warm_oso = w.slot_of[wo];
        const double warm_osz = w.sizes[wo];
        // the coordinates of the first candidate are requested now, and the half of the heap replay that does not depend on the scan's result (the
        // entry that goes after the merge, :1792-1797) runs under that round trip
        double cv0[kRomChunk / 64];
        {
            const int c0 = ncand > 0 ? s_cand[0] : st.scanned;
            const double *cc0 = w.C + static_cast<size_t>(c0 >= 0 && c0 < n2 ? c0 : st.scanned) * d;
#pragma unroll
            for (int j = 0; j < kRomChunk / 64; ++j) { const int k = lane + 64 * j; cv0[j] = cc0[k < d ? k : d - 1]; }
        }
        sel.scan_begin();
        ROM_STAMP(3);                          // heap.remove
        for (int b0 = 0; b0 < ncand; b0 += kRomBatch) {
            const int nb = ncand - b0 < kRomBatch ? ncand - b0 : kRomBatch;
            double sum = 0.0;
            for (int k0 = 0; k0 < d; k0 += kRomChunk) {
                // squares of the coordinate differences by the whole wavefront (one rounding each, as the reference's loop body) ...
                double xs[kRomChunk / 64];
                int kc[kRomChunk / 64];
#pragma unroll
                for (int j = 0; j < kRomChunk / 64; ++j) { const int k = k0 + lane + 64 * j; kc[j] = k < d ? k : d - 1; }   // clamped: every request unconditional, all in flight
                if (k0 == 0) {
#pragma unroll
                    for (int j = 0; j < kRomChunk / 64; ++j) xs[j] = xs0[j];
                } else {
#pragma unroll
                    for (int j = 0; j < kRomChunk / 64; ++j) xs[j] = cs[kc[j]];
                }
                for (int r = 0; r < nb; ++r) {
                    const double *cc = w.C + static_cast<size_t>(s_cand[b0 + r]) * d;
                    double cv[kRomChunk / 64];
                    if (b0 == 0 && k0 == 0 && r == 0) {
#pragma unroll
                        for (int j = 0; j < kRomChunk / 64; ++j) cv[j] = cv0[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < kRomChunk / 64; ++j) cv[j] = cc[kc[j]];
                    }
#pragma unroll
                    for (int j = 0; j < kRomChunk / 64; ++j) { const double diff = __dsub_rn(cv[j], xs[j]); s_t[r][lane + 64 * j] = __dmul_rn(diff, diff); }
                }
                WaveMem::wave_sync();
                // ... summed by ONE lane per candidate in the reference's order (sqeuclidean_extended, FastClusterWrapper.cpp:68-75: sequential in k)
                if (lane < nb) {
                    const int kn = d - k0 < kRomChunk ? d - k0 : kRomChunk;
                    if (kn == kRomChunk) {      // 32 values travel LDS -> registers while the previous 32 are added (the chain of additions is the floor)
                        const double2 *tp = reinterpret_cast<const double2 *>(&s_t[lane][0]);
                        double2 ta[16], tb[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) ta[q] = tp[q];
#pragma unroll
                        for (int h = 0; h < kRomChunk / 64; ++h) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) tb[q] = tp[32 * h + 16 + q];
#pragma unroll
                            for (int q = 0; q < 16; ++q) { sum = __dadd_rn(sum, ta[q].x); sum = __dadd_rn(sum, ta[q].y); }
                            if (h + 1 < kRomChunk / 64) {
#pragma unroll
                                for (int q = 0; q < 16; ++q) ta[q] = tp[32 * (h + 1) + q];
                            }
#pragma unroll
                            for (int q = 0; q < 16; ++q) { sum = __dadd_rn(sum, tb[q].x); sum = __dadd_rn(sum, tb[q].y); }
                        }
                    } else
                        for (int kk = 0; kk < kn; ++kk) sum = __dadd_rn(sum, s_t[lane][kk]);
                }
                WaveMem::wave_sync();
            }
            const bool mine = lane < nb;
            if (__builtin_amdgcn_ballot_w64(mine && sum != sum)) nan_here = true;
            if (ncand == 1) {                  // one candidate: lane 0 holds the answer
                best = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sum)), __builtin_amdgcn_readfirstlane(__double2loint(sum)));
                best_id = s_cand[0];
                if (best != best) { best = dinf(); best_id = INT_MAX; }
            } else {
                const double sv = (mine && sum == sum) ? sum : dinf();
                const double bm = wave_min(sv);
                const int bi = static_cast<int>(wave_umin((mine && sv == bm && bm < dinf()) ? static_cast<unsigned>(s_cand[b0 + lane]) : static_cast<unsigned>(INT_MAX)));
                if (lt2(bm, bi, best, best_id)) { best = bm; best_id = bi; }
            }
        }
        rom_sink(warm_onx); rom_sink(warm_opv); rom_sink(warm_oso); rom_sink(warm_osz);
    }
    ROM_STAMP(2);                              // candidates evaluated
    rom_sink(warm_pos); rom_sink(xs0[0]); rom_sink(warm_last.x); rom_sink(warm_chain.x);
#pragma unroll
    for (int q = 0; q < kWarmPer; ++q) { rom_sink(warm_top[q].x); rom_sink(warm_tree[q].x); }
    rom_sink(warm_ng); rom_sink(warm_nx); rom_sink(warm_pv); rom_sink(warm_so); rom_sink(warm_sz);
    const bool nan_flag = flag0 != 0 || nan_here;
    if (nan_flag || best_id == INT_MAX) {       // NaN distance (nan_error) / nothing to scan
        if (lane == 0) { st.done = 1; st.nan_seen = nan_flag ? 1 : 2; w.dev[0] = st; w.dev[1] = st; }
        return;
    }
#ifdef FA_ROM_PROFILE   // scan_finish's statements one by one
    if (sel.op == fa_ro::RO_NEW_ROW) {
        const int32_t created = sel.n + sel.merges - 1;
        sel.nghbr[created] = best_id;
        sel.heap.replace(sel.a, created, best);
        ROM_STAMP(4);                          // heap.replace
    } else {
        sel.nghbr[sel.a] = best_id;
        sel.heap.raise(sel.a, best);
        ROM_STAMP(5);
    }
    sel.advance();
    ROM_STAMP(6);                              // advance
#else
    sel.scan_finish(best, best_id);
#endif
    rom_prepare(st, sel, w.slot_of, w.sizes);
    st.scans = st.scans + 1;
    if (lane == 0) { w.dev[ph ^ 1] = st; if (st.done) w.dev[ph] = st; }   // the end is written to both records: every later launch of the replay returns at once
#ifdef FA_ROM_PROFILE
    ROM_STAMP(7);                              // next state
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&w.prof[i], t_seg[i]);
        atomicAdd(&w.prof[14], wall_clock64() - w_begin);
        atomicAdd(&w.prof[15], 1ULL);
    }
#endif
}

// ------------------------------------------------------------------------------ host driver
struct Layout {
    size_t state, cnt, flags, prof, c, xt, row, e2, node, sizes, z, reca, reci, recs, recp, cand, pairs, norms, m, part_vs, part_ix, total;
};

size_t rom_total_bytes(size_t N, size_t Np, size_t d);   // workspace of the matrix-filtered reference-order run (below)

Layout make_layout(size_t N, size_t Np, size_t d, size_t nblk) {
    Layout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~static_cast<size_t>(255); return at; };
    L.state = take(sizeof(AhcState) * 2);
    L.cnt = take(sizeof(WinCounters) * 4);
    L.flags = take(sizeof(int32_t) * 4);
    L.prof = take(sizeof(unsigned long long) * 16);
    L.reca = take(sizeof(RecA) * 2 * nblk);
    L.reci = take(sizeof(int4) * 2 * nblk);
    L.recs = take(sizeof(RecS) * 2 * nblk);
    L.recp = take(sizeof(RecP) * 2 * kPend * nblk);
    L.row = take(sizeof(RowSt) * Np);
    L.e2 = take(sizeof(double) * Np);
    L.node = take(sizeof(int32_t) * Np);
    L.sizes = take(sizeof(double) * 2 * N);
    L.z = take(sizeof(double) * 4 * (N > 1 ? N - 1 : 1));
    L.cand = take(sizeof(int2) * kMaxCand);
    L.pairs = take(sizeof(int4) * kMaxPairs);
    L.norms = take(sizeof(double) * Np);
    L.c = take(sizeof(double) * d * 2 * N);
    L.xt = take(sizeof(double) * d * Np);
    L.m = take(sizeof(double) * Np * Np);
    L.part_vs = take(sizeof(double2) * (Np / GT) * Np);   // per-tile row minima of the Gram start-up (0.13 % of the matrix each)
    L.part_ix = take(sizeof(int32_t) * (Np / GT) * Np);
    L.total = std::max(o, rom_total_bytes(N, Np, d));   // a run that meets an exact tie continues in reference order in the SAME workspace (no second hipMalloc of N^2 * 8 B)
    return L;
}

// d_data: device [N][d]; d_Z: device [(N-1)*4] (heights already square-rooted on return).
}  // namespace (reopened below: the next function is one of the device-level cores declared in fa_common.h)

namespace {

struct Prob {   // one linkage problem: its workspace, its copy of the device state, its outcome
    Ws w{};
    Layout L{};
    char *base = nullptr;
    size_t N = 0, Np = 0, d = 0;
    int cpt = 1;             // slots per thread of the round kernel that serves the problem: a block record covers kBlk * cpt slots, Np is a multiple of that
    const double *d_data = nullptr;
    double *d_Z = nullptr;
    int mode = FA_AHC_MODE_AUTO;
    AhcState h{};
    long long fallback = 0;
    fa_status st = FA_SUCCESS;
    bool active = true;
    bool z_on_host = false;  // d_Z is the caller's host buffer
    bool needs_ro = false;   // an exact tie at the minimum (or a window overflowing with near-ties): to be recomputed in reference order
};

void window_counter_init(WinCounters (&c)[4]) { for (auto &x : c) { x.stale_key = ~0ULL; x.ncand = 0; x.npairs = 0; } }

fa_status prob_check_shape(fa_ctx *ctx, size_t N, size_t d) {
    const size_t Np = (N + kBlk - 1) / kBlk * kBlk;
    if (Np / kBlk > static_cast<size_t>(kMaxBlocks)) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: N too large for the resident distance matrix");
    if (d * sizeof(double) > 60 * 1024) return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: dimension too large for the LDS centroid buffer");
    return FA_SUCCESS;
}

// binds the workspace at `base`, uploads the initial state and runs the start-up kernels (matrix, row minima, records, eps)
fa_status prob_setup(fa_ctx *ctx, Prob &p, char *base) {
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

    const int dev_mode = p.mode == FA_AHC_MODE_EXACT ? FA_AHC_MODE_EXACT : FA_AHC_MODE_AUTO;
    bool minima_done = false;   // the Gram tiles left per-tile row minima (ahc_gram_mfma2_t<true>): no second pass over the matrix
    hipLaunchKernelGGL(ahc_init_state, dim3(1), dim3(64), 0, ctx->stream, w, dev_mode);
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.C, p.d_data, sizeof(double) * N * d, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(ahc_init_rows, dim3((std::max(Np, 2 * N) + 255) / 256), dim3(256), 0, ctx->stream, w);
    hipLaunchKernelGGL(ahc_transpose, dim3((Np + 31) / 32, (d + 31) / 32), dim3(256), 0, ctx->stream, p.d_data, w.XT, w.N, w.Np, w.d);
    if (dev_mode == FA_AHC_MODE_AUTO) {  // Gram form on the fp64 matrix cores (approximate entries, see ahc_gram_mfma)
        double *d_norms = reinterpret_cast<double *>(base + L.norms);
        hipLaunchKernelGGL(ahc_sqnorms, dim3((w.Np + 63) / 64), dim3(256), 0, ctx->stream, w, d_norms);
        if (w.d % G2K == 0 && !fa::sw_on(fa::Sw::AHC_GRAM_V1)) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_gram_mfma2_t<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kGram2LdsBytes));
            FA_HIP_TRY(ctx, attr);
            double2 *part_vs = reinterpret_cast<double2 *>(base + L.part_vs);
            int *part_ix = reinterpret_cast<int *>(base + L.part_ix);
            hipLaunchKernelGGL(ahc_gram_mfma2_t<true>, dim3(w.Np / GT, w.Np / GT), dim3(256), kGram2LdsBytes, ctx->stream, w, d_norms, part_vs, part_ix);
            hipLaunchKernelGGL(ahc_row_minima_parts, dim3(w.Np / 64), dim3(256), 0, ctx->stream, w, part_vs, part_ix);
            minima_done = true;
        } else
            hipLaunchKernelGGL(ahc_gram_mfma, dim3(w.Np / GT, w.Np / GT), dim3(256), 0, ctx->stream, w, d_norms);
    } else {
        const int tiles = w.Np / PT;
        hipLaunchKernelGGL(ahc_pairwise, dim3(tiles, tiles), dim3(256), 0, ctx->stream, w);
    }
    if (!minima_done) hipLaunchKernelGGL(ahc_row_minima, dim3(w.Np), dim3(kBlk), 0, ctx->stream, w);
    hipLaunchKernelGGL(ahc_set_eps, dim3(1), dim3(64), 0, ctx->stream, w);
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

// rounds per replay for a problem of n points: one replay should finish a small problem (one round per merge + a few re-scans /
// window rounds) without hundreds of idle rounds behind it — at n = 50 the fixed 512-round graph cost 2.7 ms per call, five times the
// reference on a host core; large problems use the full length.  Multiple of 4 (counter rotation and parity).
inline int rounds_for(size_t n) {
    const size_t want = n + n / 8 + 8;
    const size_t r = want < static_cast<size_t>(kRoundsPerGraph) ? want : static_cast<size_t>(kRoundsPerGraph);
    return static_cast<int>((r + 3) & ~static_cast<size_t>(3));
}

struct RoundGraph {   // `rounds` rounds captured once, replayed until every problem reports done
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = false;
    int rounds = kRoundsPerGraph;
    ~RoundGraph() { if (exec) (void)hipGraphExecDestroy(exec); if (graph) (void)hipGraphDestroy(graph); }
    template <class Launch> void capture(fa_ctx *ctx, Launch &&launch, const int n_rounds) {
        ok = true;
        rounds = n_rounds;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int i = 0; i < rounds; ++i) launch(i & 3);
            if (hipStreamEndCapture(ctx->stream, &graph) != hipSuccess || !graph) ok = false;
            else if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) ok = false;
        } else ok = false;
        (void)hipGetLastError();
    }
    template <class Launch> fa_status replay(fa_ctx *ctx, Launch &&launch) {
        if (ok) FA_HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
        else for (int i = 0; i < rounds; ++i) launch(i & 3);
        return FA_SUCCESS;
    }
};

}  // namespace

namespace {

// The whole problem in the reference's selection order (see the kernels above).  d_data / d_Z: device pointers.
fa_status ro_run_device_mf(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host = false) {
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
    for (auto &e : ev) FA_HIP_TRY(ctx, hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 3; ++i) (void)hipEventDestroy(e[i]); } } evg{ev};
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], st));
    // ---- start-up: nearest lower-indexed neighbours (the reference's sums), no matrix
    FA_HIP_TRY(ctx, hipMemsetAsync(base + o_dev, 0, o_part - o_dev, st));            // RoDev, flags
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.C, d_data, sizeof(double) * N * d, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(ro_init, dim3(static_cast<unsigned>((std::max(Np, 2 * N) + 255) / 256)), dim3(256), 0, st, w);
    hipLaunchKernelGGL(ahc_transpose, dim3((Np + 31) / 32, (d + 31) / 32), dim3(256), 0, st, d_data, w.XT, w.N, w.Np, w.d);
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


// Workspace of the matrix-filtered run: the selection's arrays, then what the start-up kernels of the filter-based rounds expect (points / centroids,
// transpose, norms, the two state records their maxima go to), the matrix last.
fa_status ctx_events(fa_ctx *ctx, hipEvent_t (&ev)[3]);   // below: the three timing events a context keeps
struct RomLayout { size_t prof, dev, flags, part, part2, node, slot, sizes, key, ent, pos, ngh, next, prev, pa, pb, hs, z, state, norms, c, xt, m, total; };
RomLayout rom_layout(size_t N, size_t Np, size_t d) {
    RomLayout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~static_cast<size_t>(255); return at; };
    const size_t nblk = Np / kBlk;
    L.dev = take(sizeof(RomDev) * 2); L.flags = take(16); L.state = take(sizeof(AhcState) * 2); L.prof = take(8 * 16);
    L.part = take(sizeof(RomPart) * nblk); L.part2 = take(8 * nblk);
    L.node = take(4 * Np); L.slot = take(4 * 2 * N); L.sizes = take(8 * 2 * N); L.key = take(8 * N); L.ent = take(sizeof(fa_ro::Ent) * N); L.pos = take(4 * 2 * N);
    L.ngh = take(4 * 2 * N); L.next = take(4 * (2 * N + 1)); L.prev = take(4 * (2 * N + 1));
    L.pa = take(8 * N); L.pb = take(8 * N); L.hs = take(8 * N); L.z = take(8 * 4 * N);
    L.norms = take(8 * Np); L.c = take(8 * d * 2 * N); L.xt = take(8 * d * Np); L.m = take(8 * Np * Np);
    L.total = o;
    return L;
}

size_t rom_total_bytes(size_t N, size_t Np, size_t d) { return rom_layout(N, Np, d).total; }

// The whole problem in the reference's selection order with the matrix as the filter of its scans.  `declined` (no error recorded): the matrix cannot be
// had, or the Gram-form start-up met a non-finite entry (infinite coordinates: the sums of the matrix-free run decide what they mean) — the caller runs
// the matrix-free form instead.
fa_status rom_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host, bool &declined) {
    declined = true;
    if (N < 2 || d * sizeof(double) > 60 * 1024) return FA_SUCCESS;
    const size_t Np = (N + kBlk - 1) / kBlk * kBlk, nblk = Np / kBlk;
    if (nblk > static_cast<size_t>(kMaxBlocks)) return FA_SUCCESS;
    const RomLayout L = rom_layout(N, Np, d);
    if (fa::ws_acquire(ctx, L.total) != FA_SUCCESS) { ctx->last_error.clear(); return FA_SUCCESS; }
    char *base = static_cast<char *>(ctx->ahc_ws);
    RomWs w{};
    w.dev = reinterpret_cast<RomDev *>(base + L.dev); w.flags = reinterpret_cast<int32_t *>(base + L.flags);
    w.part = reinterpret_cast<RomPart *>(base + L.part); w.part2 = reinterpret_cast<double *>(base + L.part2);
    w.prof = reinterpret_cast<unsigned long long *>(base + L.prof);
    w.node = reinterpret_cast<int32_t *>(base + L.node); w.slot_of = reinterpret_cast<int32_t *>(base + L.slot); w.sizes = reinterpret_cast<double *>(base + L.sizes);
    w.ent = reinterpret_cast<fa_ro::Ent *>(base + L.ent); w.pos = reinterpret_cast<int32_t *>(base + L.pos);
    w.nghbr = reinterpret_cast<int32_t *>(base + L.ngh); w.next = reinterpret_cast<int32_t *>(base + L.next); w.prev = reinterpret_cast<int32_t *>(base + L.prev);
    w.pair_a = reinterpret_cast<double *>(base + L.pa); w.pair_b = reinterpret_cast<double *>(base + L.pb); w.height_sq = reinterpret_cast<double *>(base + L.hs);
    w.C = reinterpret_cast<double *>(base + L.c); w.XT = reinterpret_cast<double *>(base + L.xt); w.M = reinterpret_cast<double *>(base + L.m);
    w.N = static_cast<int32_t>(N); w.Np = static_cast<int32_t>(Np); w.d = static_cast<int32_t>(d); w.nblk = static_cast<int32_t>(nblk);
    RoWs rw{};   // the kernels shared with the matrix-free run (ro_init, ro_lower_minima_direct, ro_finish) see their own view of the same arrays
    rw.C = w.C; rw.XT = w.XT; rw.sizes = w.sizes; rw.key = reinterpret_cast<double *>(base + L.key); rw.pair_a = w.pair_a; rw.pair_b = w.pair_b; rw.height_sq = w.height_sq;
    rw.Z = reinterpret_cast<double *>(base + L.z); rw.node = w.node; rw.slot_of = w.slot_of; rw.nghbr = w.nghbr; rw.flags = w.flags;
    rw.N = w.N; rw.Np = w.Np; rw.d = w.d; rw.nblk = w.nblk;
    Ws gw{};     // ... and the Gram-form start-up of the filter-based rounds its own (its non-finite flag lands in flags[2])
    gw.state = reinterpret_cast<AhcState *>(base + L.state); gw.flags = w.flags + 2; gw.node = w.node; gw.XT = w.XT; gw.M = w.M; gw.C = w.C;
    gw.N = w.N; gw.Np = w.Np; gw.d = w.d; gw.nblk = w.nblk;
    double *d_norms = reinterpret_cast<double *>(base + L.norms);
    hipStream_t st = ctx->stream;
    hipEvent_t ev[3];
    FA_TRY(ctx_events(ctx, ev));                // the context's own three events (created once, destroyed with the context)
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], st));
    // ---- start-up: the reference's nearest lower-indexed neighbours (exact sums), and the Gram-form matrix of all pairs
    FA_HIP_TRY(ctx, hipMemsetAsync(base + L.dev, 0, L.part - L.dev, st));              // RomDev, flags, the two state records (their maxima start at 0)
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.C, d_data, sizeof(double) * N * d, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(ro_init, dim3(static_cast<unsigned>((std::max(Np, 2 * N) + 255) / 256)), dim3(256), 0, st, rw);
    hipLaunchKernelGGL(ahc_transpose, dim3((Np + 31) / 32, (d + 31) / 32), dim3(256), 0, st, d_data, w.XT, w.N, w.Np, w.d);
    const bool direct_start = fa::sw_on(fa::Sw::AHC_ROM_DIRECT_START);   // the start-up of the matrix-free run (all N^2 / 2 exact sums) for A/B
    if (direct_start) hipLaunchKernelGGL(ro_lower_minima_direct, dim3(static_cast<unsigned>((N + kRoT - 1) / kRoT)), dim3(256), 0, st, rw);
    hipLaunchKernelGGL(ahc_sqnorms, dim3((w.Np + 63) / 64), dim3(256), 0, st, gw, d_norms);
    if (w.d % G2K == 0) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_gram_mfma2_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kGram2LdsBytes));
        FA_HIP_TRY(ctx, attr);
        hipLaunchKernelGGL(ahc_gram_mfma2_t<false>, dim3(w.Np / GT, w.Np / GT), dim3(256), kGram2LdsBytes, st, gw, d_norms, static_cast<double2 *>(nullptr), static_cast<int *>(nullptr));
    } else
        hipLaunchKernelGGL(ahc_gram_mfma, dim3(w.Np / GT, w.Np / GT), dim3(256), 0, st, gw, d_norms);
    if (!direct_start) hipLaunchKernelGGL(rom_lower_minima, dim3(static_cast<unsigned>(N - 1)), dim3(kBlk), 0, st, w, gw.state, rw.key);
    FA_HIP_TRY(ctx, hipGetLastError());
    // ---- the heap over points 1 .. N-1, the list, the first pair: host (the selection logic is the same header on both sides)
    std::vector<double> key(2 * N, 0.0), pa(N, 0.0), pb(N, 0.0), hs(N, 0.0);
    std::vector<int32_t> at(N, 0), pos(2 * N, 0), ngh(2 * N, 0), next(2 * N + 1, 0), prev(2 * N + 1, 0);
    int32_t hflags[4] = {0, 0, 0, 0};
    AhcState hstate{};
    FA_HIP_TRY(ctx, hipMemcpyAsync(key.data(), rw.key, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(ngh.data(), w.nghbr, sizeof(int32_t) * N, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(hflags, w.flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(&hstate, gw.state, sizeof(hstate), hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (hflags[2]) return FA_SUCCESS;                                                  // declined: a non-finite Gram entry (the matrix-free run decides what the input means)
    if (hflags[0]) { declined = false; return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance"); }
    declined = false;
    fa_ro::Sel sel{};
    sel.heap.key = key.data(); sel.heap.at = at.data(); sel.heap.pos = pos.data();
    sel.heap.init_identity(static_cast<int32_t>(N) - 1, 1);
    sel.heap.heapify();
    sel.list.next = next.data(); sel.list.prev = prev.data();
    sel.list.init(2 * static_cast<int32_t>(N) - 1);
    sel.nghbr = ngh.data(); sel.n = static_cast<int32_t>(N); sel.merges = 0; sel.pair_a = pa.data(); sel.pair_b = pb.data(); sel.height_sq = hs.data();
    sel.advance();
    std::vector<fa_ro::Ent> ent(N);
    for (int32_t p = 0; p < sel.heap.size; ++p) { ent[p].key = key[at[p]]; ent[p].node = at[p]; ent[p].pad = 0; }
    std::vector<int32_t> slot0(2 * N, -1);
    for (size_t i = 0; i < N; ++i) slot0[i] = static_cast<int32_t>(i);
    const std::vector<double> ones(2 * N, 1.0);
    RomDev hd{};
    rom_prepare(hd, sel, slot0.data(), ones.data());
    {   // eps of ahc_set_eps + the term of d(a, b): here the pair's exact squared distance is the reference's SEQUENTIAL sum (<= (d + 2) u of it, weight
        // wa wb <= 1/4 per merge level) where the filter-based rounds sum it as a tree
        const double dmax = __builtin_bit_cast(double, hstate.dmax_bits), nmax = __builtin_bit_cast(double, hstate.nmax_bits), u = 1.1102230246251565e-16;
        hd.eps = (16.0 + 0.25 * (static_cast<double>(d) + 2.0)) * static_cast<double>(N) * u * dmax + 8.0 * (static_cast<double>(d) + 2.0) * u * nmax;
    }
    FA_HIP_TRY(ctx, hipMemcpyAsync(w.ent, ent.data(), sizeof(fa_ro::Ent) * N, hipMemcpyHostToDevice, st));
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
    // ---- one (scan, select) pair per dendrogram row, re-scan or exact re-evaluation, replayed from a graph until the device reports the end
    const size_t lds = sizeof(double) * d;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rom_scan), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    auto launch = [&](const int ph4) {
        hipLaunchKernelGGL(rom_scan, dim3(w.nblk + 1), dim3(kBlk), lds, st, w, ph4 & 1);
        hipLaunchKernelGGL(rom_select, dim3(1 + w.nblk), dim3(64), 0, st, w, ph4 & 1);
    };
    RoundGraph rg;
    rg.capture(ctx, launch, static_cast<int>(std::min<size_t>(256, (N + 3) & ~static_cast<size_t>(3))));
    const long long max_replays = 16 + 16 * static_cast<long long>(N) / rg.rounds;   // rows + re-scans + exact re-evaluations
    for (long long it = 0; it < max_replays && !hd.done; ++it) {
        FA_TRY(rg.replay(ctx, launch));
        FA_HIP_TRY(ctx, hipMemcpyAsync(&hd, w.dev, sizeof(hd), hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    if (hd.nan_seen == 1) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance");
    if (!hd.done || hd.nan_seen || hd.merges != static_cast<int32_t>(N) - 1) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: reference-order run stopped at row %d", hd.merges);
    hipLaunchKernelGGL(ro_finish, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, st, rw);
    FA_HIP_TRY(ctx, hipMemcpyAsync(d_Z, rw.Z, sizeof(double) * 4 * (N - 1), z_on_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    FA_HIP_TRY(ctx, hipEventRecord(ev[2], st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));
#ifdef FA_ROM_PROFILE
    {
        unsigned long long hp[16];
        (void)hipMemcpy(hp, w.prof, sizeof(hp), hipMemcpyDeviceToHost);
        const double n = hp[15] ? static_cast<double>(hp[15]) : 1.0;
        fprintf(stderr, "rom_select profile (clocks per selection, %llu selections; wall 100 MHz ticks %.1f): first trip %.0f | candidates %.0f | evaluation %.0f | heap.remove %.0f | heap.replace %.0f | raise %.0f | advance %.0f | next state %.0f\n",
                hp[15], hp[14] / n, hp[0] / n, hp[1] / n, hp[2] / n, hp[3] / n, hp[4] / n, hp[5] / n, hp[6] / n, hp[7] / n);
    }
#endif
    if (fa::sw(fa::Sw::AHC_DEBUG))
        fprintf(stderr, "ahc (reference order, matrix filter): N %zu scans %lld exact re-evaluations %lld candidates %lld eps %.3e\n", N, hd.scans, hd.exact_scans, hd.cands, hd.eps);
    if (stats) {
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        stats->merges = hd.merges; stats->rounds += hd.scans + hd.exact_scans; if (!stats->reference_order) stats->reference_order = 1;
        stats->rescans += hd.exact_scans;   // rows whose candidates were too many for one wavefront: scanned again with exact sums
        stats->init_ms += t01; stats->merge_ms += t12; stats->total_ms += t01 + t12;
    }
    return FA_SUCCESS;
}

// The reference-order run: through the matrix filter when the workspace is to be had, matrix-free (O(N d) memory, O(A d) sums per row) when not.
fa_status ro_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host = false) {
    if (!fa::sw_on(fa::Sw::AHC_RO_NO_MATRIX)) {
        bool declined = false;
        const fa_status st = rom_run_device(ctx, d_data, N, d, d_Z, stats, z_on_host, declined);
        if (!declined) return st;
    }
    return ro_run_device_mf(ctx, d_data, N, d, d_Z, stats, z_on_host);
}

}  // namespace

namespace {
struct CachedGraph {   // the round launches of one problem shape, kept in the context between calls
    RoundGraph rg;
    const void *base = nullptr;
    size_t N = 0, d = 0;
    int cpt = 1;
    int grid_y = 0, kernel = 0;   // uniform batches: problems in the grid and which build of the round serves them
};
void cached_graph_free(void *p) { delete static_cast<CachedGraph *>(p); }
fa_status ctx_events(fa_ctx *ctx, hipEvent_t (&ev)[3]) {
    for (int i = 0; i < 3; ++i) {
        if (!ctx->ahc_ev[i]) FA_HIP_TRY(ctx, hipEventCreate(&ctx->ahc_ev[i]));
        ev[i] = ctx->ahc_ev[i];
    }
    return FA_SUCCESS;
}
}  // namespace

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
    const size_t lds = sizeof(double) * d;

    hipEvent_t ev[3];
    FA_TRY(ctx_events(ctx, ev));                // created once per context
    FA_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    FA_TRY(prob_setup(ctx, p, static_cast<char *>(ctx->ahc_ws)));
    FA_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));

    const Ws w = p.w;
    const bool env_big = fa::sw_on(fa::Sw::AHC_ROUND_BIG);
    const bool big = w.nblk > (4 / p.cpt) * 64 || env_big;   // more than 65 536 points (four block records per lane at one slot per thread): the kernel with the many-record reduction
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ahc_round_t<false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    }
    auto off_of = [&](const void *p) { return static_cast<unsigned>(static_cast<const char *>(p) - reinterpret_cast<const char *>(w.state)); };   // small arrays: within 4 GB of the state (make_layout puts the matrix last)
    const unsigned o_row = off_of(w.row), o_node = off_of(w.node), o_e2 = off_of(w.e2), o_flags = off_of(w.flags);
#define FA_AHC_ROUND_LAUNCH(BIG_, CPT_) hipLaunchKernelGGL((ahc_round_t<false, BIG_, CPT_>), dim3(w.nblk), dim3(kBlk), lds, ctx->stream, ph, w.nblk, w.state, w.recA, w.recI, w.recP, \
                                                          o_row, o_node, o_e2, o_flags, w, static_cast<const Ws *>(nullptr), static_cast<const int2 *>(nullptr))
    auto launch = [&](const int ph) {
        if (p.cpt == 4) { if (big) FA_AHC_ROUND_LAUNCH(true, 4); else FA_AHC_ROUND_LAUNCH(false, 4); }
        else if (p.cpt == 2) { if (big) FA_AHC_ROUND_LAUNCH(true, 2); else FA_AHC_ROUND_LAUNCH(false, 2); }
        else { if (big) FA_AHC_ROUND_LAUNCH(true, 1); else FA_AHC_ROUND_LAUNCH(false, 1); }
    };
#undef FA_AHC_ROUND_LAUNCH
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
    FA_TRY(prob_finish(ctx, p));
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
        return ro_run_device(ctx, d_data, N, d, d_Z, stats, z_on_host);
    }
#ifdef FA_AHC_PROFILE
    {
        unsigned long long hp[16];
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
        p.st = ro_run_device(ctx, p.d_data, p.N, p.d, p.d_Z, stats ? &stats[k] : nullptr);
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
size_t uniform_stride(int count, size_t Nmax, size_t d) {
    const int cpt = uniform_cpt(count, Nmax);
    const size_t cols = static_cast<size_t>(kBlk) * cpt, Np = (Nmax + cols - 1) / cols * cols;
    return (make_layout(Nmax, Np, d, Np / cols).total + 4095) & ~static_cast<size_t>(4095);
}

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
        if (cpt == 4) hipLaunchKernelGGL(ahc_round_uni_c4, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
        else if (cpt == 2) hipLaunchKernelGGL(ahc_round_uni_c2, grid, dim3(kBlk), lds, ctx->stream, a0, stride_pages, w0.state, w0.recA, w0.recI, w0.recP, o_row, o_node, o_e2, o_flags, w0);
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
        p.st = ro_run_device(ctx, p.d_data, p.N, p.d, p.d_Z, stats ? &stats[ord[j]] : nullptr);
        if (statuses) statuses[ord[j]] = p.st;
        if (p.st != FA_SUCCESS && worst == FA_SUCCESS) worst = p.st;
    }
    *completed = true;
    return worst;
}

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
int uniform_groups(int count, const size_t *n) {
    if (const char *e = fa::sw(fa::Sw::AHC_UNI_GROUPS)) { const int v = atoi(e); if (v >= 1 && v <= 4) return std::min(v, count / 2 > 0 ? count / 2 : 1); }
    size_t lo = SIZE_MAX;
    for (int k = 0; k < count; ++k) lo = std::min(lo, n[k]);
    // six long recordings, or eight medium ones (chains of >= 4 096 rounds: the second stream's thread + graph capture, ~2 ms, must be worth it)
    return (count >= 6 && lo >= kInFlightMinN) || (count >= 8 && lo >= kUniGroupsMinN) ? 2 : 1;
}

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

namespace {

// normalizeFeatures (AHCClustering.swift:70-105): one thread per row, the reference's sequential sum of squares (this file
// is compiled with -ffp-contract=off), scale = norm > 0 ? 1 / sqrt(norm) : 0 — bit-identical to the host loop of fa_ahc_cluster.
__global__ void ahc_normalize_rows(const double *__restrict__ x, double *__restrict__ out, int64_t n, int d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *row = x + i * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss += row[k] * row[k];
    const double scale = ss > 0 ? 1.0 / sqrt(ss) : 0.0;
    for (int k = 0; k < d; ++k) out[i * d + k] = row[k] * scale;
}

}  // namespace

fa_status fa::ahc_normalize_dev(fa_ctx *ctx, const double *d_x, double *d_out, int64_t n, int32_t d) {
    if (n <= 0) return FA_SUCCESS;
    hipLaunchKernelGGL(ahc_normalize_rows, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(64), 0, ctx->stream, d_x, d_out, n, d);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

namespace {

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
        if (device_pointers) return fa::ahc_run_device(ctx, data, n, d, dendrogram, mode, stats);
        // input staged in the context's grow-only scratch (no hipMalloc / hipFree per call), dendrogram copied straight from the workspace
        FA_TRY(fa::ensure_scratch(ctx, sizeof(double) * n * d));
        FA_HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch, data, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream));
        return fa::ahc_run_device(ctx, static_cast<const double *>(ctx->scratch), n, d, dendrogram, mode, stats, /*z_on_host*/ true);
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FA_RUNTIME_ERROR;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

// `count` independent linkage problems (recordings) of dimension d in one call: their serial merge chains advance together
// (one launch = one round of every unfinished problem).  data[k]: n[k] x d row-major, dendrograms[k]: (n[k] - 1) x 4 — HOST
// pointers unless device_pointers != 0 (the two pointer ARRAYS are always host arrays).  statuses[k] (nullable) carries the
// per-problem status of the reference contract (n == 0 or 1 -> SUCCESS, nothing written); the return value is the first failure.
fa_status fa_ahc_linkage_batch(fa_ctx *ctx, int32_t count, const double *const *data, const size_t *n, size_t d, double *const *dendrograms,
                               int32_t mode, int32_t device_pointers, fa_ahc_stats *stats, int32_t *statuses) {
    if (!ctx || count < 0 || (count > 0 && (!data || !n || !dendrograms))) return FA_INVALID_ARGUMENT;
    if (count == 0) return FA_SUCCESS;
    try {
        fa::DeviceGuard guard(ctx->device);
        std::vector<fa_status> st(count, FA_SUCCESS);
        std::vector<const double *> d_in(count, nullptr);
        std::vector<double *> d_z(count, nullptr);
        std::vector<size_t> nn(count, 0);
        std::vector<fa::DevBuf> bufs(static_cast<size_t>(2) * count);
        for (int k = 0; k < count; ++k) {
            bool trivial;
            st[k] = linkage_checks(data[k], n[k], d, dendrograms[k], n[k] > 1 ? (n[k] - 1) * 4 : 0, &trivial);
            if (st[k] != FA_SUCCESS || trivial) continue;
            nn[k] = n[k];
            if (device_pointers) { d_in[k] = data[k]; d_z[k] = dendrograms[k]; continue; }
            if (bufs[2 * k].alloc(sizeof(double) * n[k] * d) != hipSuccess || bufs[2 * k + 1].alloc(sizeof(double) * 4 * (n[k] - 1)) != hipSuccess) {
                (void)hipGetLastError();
                st[k] = FA_ALLOCATION_FAILURE; nn[k] = 0;
                continue;
            }
            FA_HIP_TRY(ctx, hipMemcpyAsync(bufs[2 * k].p, data[k], sizeof(double) * n[k] * d, hipMemcpyHostToDevice, ctx->stream));
            d_in[k] = bufs[2 * k].as<double>(); d_z[k] = bufs[2 * k + 1].as<double>();
        }
        std::vector<fa_status> run(count, FA_SUCCESS);
        (void)fa::ahc_run_device_batch(ctx, count, d_in.data(), nn.data(), d, d_z.data(), mode, stats, run.data());
        fa_status first = FA_SUCCESS;
        for (int k = 0; k < count; ++k) {
            if (st[k] == FA_SUCCESS && nn[k] >= 2) st[k] = run[k];
            if (st[k] == FA_SUCCESS && nn[k] >= 2 && !device_pointers)
                FA_HIP_TRY(ctx, hipMemcpyAsync(dendrograms[k], d_z[k], sizeof(double) * 4 * (nn[k] - 1), hipMemcpyDeviceToHost, ctx->stream));
            if (statuses) statuses[k] = st[k];
            if (st[k] != FA_SUCCESS && first == FA_SUCCESS) first = st[k];
        }
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return first;
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
        // re-entrant from any thread: every call borrows one context of the default device set (pool.hip) for its duration, so
        // concurrent callers run on different GPUs (FLUIDAUDIO_HIP_DEVICES) and queue only when all of them are taken
        fa_pool *pool = nullptr;
        const fa_status ps = fa::default_pool(&pool);
        if (ps != FA_SUCCESS) return static_cast<fastcluster_wrapper_status>(ps == FA_ALLOCATION_FAILURE ? ps : FA_RUNTIME_ERROR);
        fa_ctx *ctx = nullptr;
        if (fa_pool_acquire(pool, &ctx) != FA_SUCCESS) return FASTCLUSTER_WRAPPER_RUNTIME_ERROR;
        struct Release { fa_pool *p; fa_ctx *c; ~Release() { fa_pool_release(p, c); } } release{pool, ctx};
        return static_cast<fastcluster_wrapper_status>(
            fa_ahc_linkage(ctx, data, pointCount, dimension, dendrogramOut, dendrogramLength, FA_AHC_MODE_AUTO, 0, nullptr));
    } catch (const std::bad_alloc &) {
        return FASTCLUSTER_WRAPPER_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FASTCLUSTER_WRAPPER_RUNTIME_ERROR;
    } catch (...) {
        return FASTCLUSTER_WRAPPER_UNKNOWN_ERROR;
    }
}

fa_status fa_ctx_reserve(fa_ctx *ctx, size_t n_max, size_t d, int32_t recordings) {
    if (!ctx || d == 0 || recordings < 1) return FA_INVALID_ARGUMENT;
    if (n_max < 2) return FA_SUCCESS;
    return fa::no_throw(ctx, "fa_ctx_reserve", [&]() -> fa_status {
        fa::DeviceGuard guard(ctx->device);
        FA_TRY(prob_check_shape(ctx, n_max, d));
        if (recordings == 1) {
            const size_t Np = (n_max + kBlk - 1) / kBlk * kBlk;
            fa::WsUse use(ctx);
            return fa::ws_acquire(ctx, make_layout(n_max, Np, d, Np / kBlk).total);
        }
        // The reservation follows the dispatch of a batch of `recordings` problems of n_max points (run_device_batch_impl): six or more long recordings
        // run as two uniform batches side by side, the second on a helper context with a workspace of its OWN — reserved here as well, so that the
        // first request pays no hipMalloc on either (until round 5 everything was reserved on the caller's context: the helper still allocated inside
        // the first request, and the two together held ~1.5 x the need).  The slot size is that of the round kernel the batch will run with.
        std::vector<size_t> n(static_cast<size_t>(recordings), n_max);
        const bool capped = ctx->ws_cap != static_cast<size_t>(-1);
        int groups = !capped && uniform_eligible(recordings, n.data(), FA_AHC_MODE_AUTO) ? uniform_groups(recordings, n.data()) : 1;
        for (int g = 1; g < groups; ++g) {
            fa_ctx *&h = ctx->helpers[g - 1];
            if (!h) {
                if (fa_ctx_create(ctx->device, nullptr, &h) != FA_SUCCESS) { h = nullptr; groups = g; break; }
                h->ws_limit = ctx->ws_limit;
                h->ws_cap = ctx->ws_cap;
            }
        }
        for (int g = 0; g < groups; ++g) {
            const int m = static_cast<int>(static_cast<long long>(recordings) * (g + 1) / groups - static_cast<long long>(recordings) * g / groups);
            fa_ctx *c = g == 0 ? ctx : ctx->helpers[g - 1];
            fa::WsUse use(c);
            const fa_status st = fa::ws_acquire(c, uniform_stride(m, n_max, d) * static_cast<size_t>(m));
            if (st != FA_SUCCESS) { if (c != ctx) ctx->last_error = c->last_error; return st; }
        }
        return FA_SUCCESS;
    });
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
        std::vector<char> merged(total, 0);
        for (size_t r = 0; r + 1 < n; ++r) {
            // The reference only ever cuts what its own wrapper wrote (AHCClustering.swift:40-58); a C caller can hand over anything.  The
            // children of row r must be two different nodes that exist when it is formed (leaves, or rows < r) and were not merged before:
            // anything else is an out-of-bounds read or an endless walk below.
            const double a = z[4 * r], b = z[4 * r + 1], limit = static_cast<double>(n + r);
            if (!(a >= 0.0 && a < limit && b >= 0.0 && b < limit) || a != std::floor(a) || b != std::floor(b) || a == b) return FA_INVALID_ARGUMENT;
            const size_t ia = static_cast<size_t>(a), ib = static_cast<size_t>(b);
            if (merged[ia] || merged[ib]) return FA_INVALID_ARGUMENT;
            merged[ia] = merged[ib] = 1;
            left[n + r] = static_cast<int64_t>(ia);
            right[n + r] = static_cast<int64_t>(ib);
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

// ---------------------------------------------------------------------------------------------------------------------
// Row minima of a SLAB of the pairwise distance matrix: for rows [row0, row1) the nearest other point among all n (the
// reference's distance: sequential fp64 sum of squared differences, FastClusterWrapper.cpp:45-52; lowest index on ties).
// This is the start-up of the linkage (fastcluster_internal.hpp:1653-1678 builds the same nearest-neighbour table) in a form
// that shards by rows across GPUs (all-gather X, per-rank slab, gather (min, idx)); fluidaudio_amd/sharding.py
// drives it.  64 x 64 output tile per workgroup step, 4 x 4 per thread, operands staged k-major in LDS.
namespace {

constexpr int kSlabT = 64, kSlabK = 16;

__global__ __launch_bounds__(256) void slab_row_minima_kernel(const double *__restrict__ x, int n, int d, int row0, int row1, double *__restrict__ out_min,
                                                              int32_t *__restrict__ out_arg) {
    __shared__ double sa[kSlabK][kSlabT + 1], sb[kSlabK][kSlabT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // tx: column quad, ty: row quad
    const int i0 = row0 + blockIdx.x * kSlabT;
    double best[4];
    int arg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { best[r] = __longlong_as_double(0x7ff0000000000000LL); arg[r] = -1; }
    for (int j0 = 0; j0 < n; j0 += kSlabT) {
        double acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
        for (int k0 = 0; k0 < d; k0 += kSlabK) {
            for (int e = tid; e < kSlabT * kSlabK; e += 256) {   // 64 rows x 16 dims of both operands
                const int rr = e / kSlabK, kk = e % kSlabK;
                const int gi = i0 + rr, gj = j0 + rr, gk = k0 + kk;
                sa[kk][rr] = gi < row1 && gk < d ? x[static_cast<size_t>(gi) * d + gk] : 0.0;
                sb[kk][rr] = gj < n && gk < d ? x[static_cast<size_t>(gj) * d + gk] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < kSlabK; ++kk) {   // ascending k, one rounding per operation: the reference's sum
                double a[4], b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = sa[kk][4 * ty + r]; b[r] = sb[kk][4 * tx + r]; }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const double df = a[r] - b[c]; acc[r][c] = acc[r][c] + df * df; }
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gi = i0 + 4 * ty + r, gj = j0 + 4 * tx + c;
                if (gi < row1 && gj < n && gj != gi && (acc[r][c] < best[r] || (acc[r][c] == best[r] && gj < arg[r]))) { best[r] = acc[r][c]; arg[r] = gj; }
            }
    }
    // the 16 threads of a row quad (tx = 0..15, consecutive lanes) combine: lowest value, then lowest index
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = best[r];
        int a = arg[r];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, 16);
            const int oa = __shfl_xor(a, off, 16);
            if (oa >= 0 && (a < 0 || ov < v || (ov == v && oa < a))) { v = ov; a = oa; }
        }
        const int gi = i0 + 4 * ty + r;
        if (tx == 0 && gi < row1) { out_min[gi - row0] = v; out_arg[gi - row0] = a; }
    }
}

}  // namespace

extern "C" fa_status fa_ahc_row_minima(fa_ctx *ctx, const double *x, size_t n, size_t d, size_t row0, size_t row1, double *mins, int32_t *args,
                                       int32_t device_pointers) {
    if (!ctx || !x || !mins || !args) return FA_INVALID_ARGUMENT;
    if (row0 > row1 || row1 > n || d == 0 || n > static_cast<size_t>(INT32_MAX) || d > static_cast<size_t>(INT32_MAX)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "row minima: bad range");
    if (row0 == row1) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    const size_t rows = row1 - row0;
    fa::DevBuf bx, bm, ba;
    const double *d_x = x;
    double *d_m = mins;
    int32_t *d_a = args;
    if (!device_pointers) {
        if (bx.alloc(sizeof(double) * n * d) != hipSuccess || bm.alloc(sizeof(double) * rows) != hipSuccess || ba.alloc(sizeof(int32_t) * rows) != hipSuccess) {
            (void)hipGetLastError();
            return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "row minima: device allocation failed");
        }
        FA_HIP_TRY(ctx, hipMemcpyAsync(bx.p, x, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream));
        d_x = bx.as<double>(); d_m = bm.as<double>(); d_a = ba.as<int32_t>();
    }
    hipLaunchKernelGGL(slab_row_minima_kernel, dim3(static_cast<unsigned>((rows + kSlabT - 1) / kSlabT)), dim3(256), 0, ctx->stream, d_x, static_cast<int>(n),
                       static_cast<int>(d), static_cast<int>(row0), static_cast<int>(row1), d_m, d_a);
    FA_HIP_TRY(ctx, hipGetLastError());
    if (!device_pointers) {
        FA_HIP_TRY(ctx, hipMemcpyAsync(mins, d_m, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipMemcpyAsync(args, d_a, sizeof(int32_t) * rows, hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FA_SUCCESS;
}
