// ahc_ws.h — what the translation units of the linkage share: constants, the device state and block records, the workspace descriptor (Ws) and
// its layout in the context's cached allocation (Layout / make_layout: THE one place that says where an array lives), wave helpers (DPP),
// the host-side record of a problem (Prob), the graph of round launches, and the functions one unit offers the others.
//   ahc_startup.hip   start-up kernels: transpose, Gram-form matrix on the fp64 matrix cores (+ per-tile row minima), exact pairwise matrix, eps
//   ahc_round_body.h  the round (block records, decision, merge / re-scan / window phases) as a template over the launch forms
//   ahc_rounds.hip    one problem: entry kernels, set-up / replay / finish, fa::ahc_run_device
//   ahc_batch.hip     several problems per launch: uniform-layout batches, argument-table batches, groups side by side, fa::ahc_run_device_batch
//   ahc_ro.hip        the reference's selection order, matrix-free (O(N d) memory)
//   ahc_rom.hip       the reference's selection order with the matrix as the filter of its scans
//   ahc_api.hip       normalisation, the C entries, the dendrogram cut, the shardable nearest-neighbour table
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "fa_common.h"
#include "ahc_reforder.h"

namespace fa_ahc {

#ifndef FA_AHC_SPECULATE
#define FA_AHC_SPECULATE 1   // the operands of the presumptive merge are requested before the decision is complete (0: after it, as in round 2)
#endif
#ifndef FA_AHC_BLK
#define FA_AHC_BLK 256
#endif
constexpr int kBlk = FA_AHC_BLK;   // rows per block record == threads per round workgroup (512 measured: see profiles/r03_ahc_variants.txt)
constexpr int kWaves = kBlk / 64;
constexpr int kMaxBlocks = 768;    // N <= 196 608 (N^2 * 8 B = 288 GB is reached at N ~ 190 000)
#ifndef FA_AHC_ROUNDS_PER_GRAPH
#define FA_AHC_ROUNDS_PER_GRAPH 512
#endif
constexpr int kRoundsPerGraph = FA_AHC_ROUNDS_PER_GRAPH;  // multiple of 4 (counter rotation) and of 2 (parity)
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
// The PACKED record (round 6), behind recI in the same array: everything the first reduction wants of a block in ONE 16-byte element — v1 and, in the other
// 64 bits, the row's slot within its block (10 bits), the window count saturated at 3 (2), the neighbour slot + 1 (17), the two node ids (17 each).  Holds
// while slots <= 65 536 and node ids < 131 072, i.e. for every problem the register path of the first reduction serves; larger ones read recA / recI.
__device__ __forceinline__ int4 rec_pack(const double v1, const int cnt, const int local, const int q1, const int nr1, const int nq1) {
    const unsigned c3 = cnt > 3 ? 3u : static_cast<unsigned>(cnt), q = static_cast<unsigned>(q1 + 1) & 0x1ffffu, nq = q1 >= 0 ? static_cast<unsigned>(nq1) & 0x1ffffu : 0u;
    const unsigned z = (static_cast<unsigned>(local) & 1023u) | (c3 << 10) | (q << 12) | ((nq >> 15) << 29);
    const unsigned wv = (static_cast<unsigned>(nr1) & 0x1ffffu) | ((nq & 0x7fffu) << 17);
    return make_int4(__double2loint(v1), __double2hiint(v1), static_cast<int>(z), static_cast<int>(wv));
}
__device__ __forceinline__ void rec_unpack(const unsigned z, const unsigned wv, int &local, int &q1, int &nr1, int &nq1) {
    local = static_cast<int>(z & 1023u);
    q1 = static_cast<int>((z >> 12) & 0x1ffffu) - 1;
    nr1 = static_cast<int>(wv & 0x1ffffu);
    nq1 = q1 >= 0 ? static_cast<int>((wv >> 17) | (((z >> 29) & 3u) << 15)) : -1;
}

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

constexpr int GT = 128;   // tile edge of the Gram kernels (ahc_startup.hip); the per-tile row minima are laid out by it
// ------------------------------------------------------------------------------ host driver
struct Layout {
    size_t state, cnt, flags, prof, c, xt, row, e2, node, sizes, z, reca, reci, recs, recp, cand, pairs, norms, m, part_vs, part_ix, total;
};

size_t rom_total_bytes(size_t N, size_t Np, size_t d);   // workspace of the matrix-filtered reference-order run (ahc_rom.hip)

// The arrays of the filter-based rounds.  `total` here is their end; make_layout() below adds what the reference-order run keeps BEHIND them (it shares
// these arrays — state, flags, node, sizes, centroids, transpose, norms, matrix, dendrogram — so that a run in reference order can hand its problem back to
// the rounds in place: ahc_rom.hip, prob_adopt).
inline Layout make_layout_core(size_t N, size_t Np, size_t d, size_t nblk) {
    Layout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~static_cast<size_t>(255); return at; };
    L.state = take(sizeof(AhcState) * 2);
    L.cnt = take(sizeof(WinCounters) * 4);
    L.flags = take(sizeof(int32_t) * 4);
    L.prof = take(sizeof(unsigned long long) * 16);
    L.reca = take(sizeof(RecA) * 2 * nblk);
    L.reci = take(sizeof(int4) * 4 * nblk);   // [2][nblk] recI, then [2][nblk] the packed records (rec_pack)
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
    L.total = o;
    return L;
}
inline Layout make_layout(size_t N, size_t Np, size_t d, size_t nblk) {
    Layout L = make_layout_core(N, Np, d, nblk);
    L.total = std::max(L.total, rom_total_bytes(N, Np, d));   // a run that meets an exact tie continues in reference order in the SAME workspace (no second hipMalloc of N^2 * 8 B)
    return L;
}

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


// ---- the reference-order runs: device records shared by ahc_ro.hip and ahc_rom.hip
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


// ---- what the units offer each other
// ahc_startup.hip (everything enqueued on `st`; no synchronisation)
void startup_filter(hipStream_t st, const Ws &w, const Layout &L, char *base, int dev_mode, const double *d_data, size_t N, size_t Np, size_t d);   // state, rows, transpose, matrix, row minima, eps
void startup_transpose(hipStream_t st, const double *d_data, double *XT, int N, int Np, int d);
fa_status startup_gram(fa_ctx *ctx, hipStream_t st, const Ws &gw, double *d_norms);   // norms + Gram-form matrix of gw (no row minima): the matrix-filtered reference-order run
// ahc_rounds.hip
void window_counter_init(WinCounters (&c)[4]);
fa_status prob_check_shape(fa_ctx *ctx, size_t N, size_t d);
void prob_bind(Prob &p, char *base);                      // p.w = the arrays of p.L at `base` (no device work)
fa_status prob_setup(fa_ctx *ctx, Prob &p, char *base);   // prob_bind + the start-up kernels
// The rounds of a problem whose workspace is bound and whose device state is ready (prob_setup, or prob_adopt): replays of the round graph until the device
// reports the end or halts, then the exact heights and the dendrogram to p.d_Z — unless p.needs_ro (an exact tie: the caller recomputes in reference order).
fa_status prob_run_rounds(fa_ctx *ctx, Prob &p);
// Adopt a clustering in progress (the reference-order run hands its problem back, ahc_rom.hip): `merges` merges are done — node[], sizes[], centroids and
// a matrix that holds BOTH copies of every live pair among nodes below N + merges - 1 (the newest node's row is valid, its column may not be) are in
// place, the merged pairs (node ids) in pair_a / pair_b.  Builds the row states, the state record, the block records and the first `merges` dendrogram rows.
fa_status prob_adopt(fa_ctx *ctx, Prob &p, int merges, double eps, const double *pair_a, const double *pair_b);
fa_status prob_after_replay(fa_ctx *ctx, Prob &p);
fa_status prob_finish(fa_ctx *ctx, Prob &p);
struct CachedGraph {   // the round launches of one problem shape, kept in the context between calls
    RoundGraph rg;
    const void *base = nullptr;
    size_t N = 0, d = 0;
    int cpt = 1;
    int grid_y = 0, kernel = 0;   // uniform batches: problems in the grid and which build of the round serves them
};
void cached_graph_free(void *p);
fa_status ctx_events(fa_ctx *ctx, hipEvent_t (&ev)[3]);   // the three timing events a context keeps
// ahc_ro.hip / ahc_rom.hip
fa_status ro_run_device_mf(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host = false);
// may_hand_over: the caller is AUTO's tie route — once the ties have stopped the rest of the problem may go back to the filter-based rounds (ahc_rom.hip)
// matrix_ready: the caller's filter-based attempt halted before its first merge in this very workspace (one slot per thread, Gram-form start-up): transpose, norms,
// matrix and the start-up's maxima are in place — the reference-order start-up does not build them again
fa_status ro_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host = false, bool may_hand_over = false,
                        bool matrix_ready = false);
size_t rom_total_bytes(size_t N, size_t Np, size_t d);   // workspace of the matrix-filtered reference-order run
void ro_launch_init(hipStream_t st, const RoWs &w, size_t threads);
void ro_launch_lower_minima_direct(hipStream_t st, const RoWs &w);
void ro_launch_finish(hipStream_t st, const RoWs &w);
// ahc_batch.hip
bool uniform_eligible(int count, const size_t *n, int mode);
int uniform_groups(int count, const size_t *n);                  // uniform batches side by side a dispatch of `count` problems uses
size_t uniform_stride(int count, size_t Nmax, size_t d);         // bytes per problem slot of a uniform batch

}  // namespace fa_ahc
