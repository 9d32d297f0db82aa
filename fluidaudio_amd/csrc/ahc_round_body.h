// ahc_round_body.h — the round of the filter-based linkage: block records, the decision every workgroup reaches on its own, the merge / re-scan /
// window phases; a template over the launch forms (ahc_rounds.hip: one problem; ahc_batch.hip: several per launch).  See ahc_ws.h for the layout.
#pragma once
#include "ahc_ws.h"

namespace fa_ahc {
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
// BRANCH (one slot per thread, single chain): the request sits behind its condition — a column that wants nothing requests nothing.  Measured both ways in
// round 6 (profiles/r06_round_uncond.txt): the single chain of 43 200 points 4.95 us per round this way against 5.05 without the branch, the launches
// over several problems the other way round (K = 8 x 8 h: 159 against 162 - 164 audio-hours/s).
template <int CPT, bool BRANCH = false>
__device__ __forceinline__ void pair_entries(const double *M, const int Np, const int r, const int nr, const int x0, const int (&nx)[CPT], const int sym_limit,
                                             const int skip0, const int skip1, double (&out)[CPT]) {
    if constexpr (CPT == 1 && BRANCH) {
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
        w.recI[2 * static_cast<size_t>(w.nblk) + o1] = rec_pack(ra.v1, ra.cnt, any ? best.a - blk * kBlk * CPT : 0, any ? best.b : -1, any ? best.c : 0, any ? best.d : 0);
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
template <bool N_IN_STATE = false, bool BIG = true, int CPT = 1, int KC = 4 / CPT>
__device__ __forceinline__ void ahc_round_body(const Ws w_in, const int blk, const int ph /* round index & 3 */, const size_t late_shift = 0) {
    static_assert(CPT == 1 || kPiggy == 0, "piggy-backed re-scans were only ever built for one slot per thread");
    constexpr int kCols = kBlk * CPT;
    // Requests of the round's second memory round trip: behind their conditions (single chain) or unconditional with dropped values (launches over several
    // problems).  A conditional request costs the compiler its count of the in-order memory counter — behind the join it no longer knows how many requests are
    // younger than an older one and waits for all of them — so the unconditional form lets the centroid arithmetic run under the latency of the two cold matrix
    // entries; it also requests for dead columns and for rounds that merge nothing.  Measured (profiles/r06_round_uncond.txt): K = 8 x 8 h 159 -> 162 - 164
    // audio-hours/s unconditional, the single chain of 43 200 points 4.95 -> 5.05 us per round: each form where it wins.
    constexpr bool kUncond = N_IN_STATE, kEntryBranch = !N_IN_STATE;
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
    // kC4 = KC: block records per lane held in registers (N <= 64 lanes x KC x kBlk x CPT; at most 65 536 slots); beyond: the generic path.  A request for a
    // record the lane does not own is not free (round 6: the four wavefronts of a workgroup share the CU's one address unit — at 43 200 points, three
    // records per lane, the fourth pair of requests cost 0.12 us of a 5.1 us round), so the host picks the instance with KC = the records a lane owns.
    constexpr int kC4 = KC;
    static_assert(KC >= 1 && KC * CPT <= 4, "at most 65 536 slots through the register path");
    int4 q0[kC4], q1[kC4];
    bool qok[kC4];
    {
        // wave 0 reads the PACKED records (rec_pack, ahc_ws.h: one request per record instead of two — 5.07 -> 4.96 us per round at 43 200 points); a second
        // request per record only where a row wave finishes two rows
        const void *b0 = wave == 0 ? static_cast<const void *>(w.recI + 2 * static_cast<size_t>(nblk) + ro) : static_cast<const void *>(w.recP + (static_cast<size_t>(par) * kPend + pk0) * nblk);
        const void *b1 = wave == 0 ? static_cast<const void *>(w.recI + ro) : static_cast<const void *>(w.recP + (static_cast<size_t>(par) * kPend + pk1) * nblk);
#pragma unroll
        for (int j = 0; j < kC4; ++j) {
            const int i = lane * c + j;
            qok[j] = c <= kC4 && j < c && i < nblk && (wave == 0 || row_wave);
            const size_t ii = qok[j] ? i : 0;
            q0[j] = rec16(b0, ii);
            if constexpr (kPend > 1) q1[j] = rec16(b1, ii); else q1[j] = q0[j];
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
    auto reduce_minimum_packed = [&](auto cc, const int4 *rx, const bool *ok) {
        constexpr int C = decltype(cc)::value;
        double va[C], key = dinf();
        int ca[C], bz = 0, bw = 0, bi = 0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            va[j] = ok[j] ? rec_f64(rx[j]) : dinf();
            ca[j] = ok[j] ? (rx[j].z >> 10) & 3 : 0;
            const bool better = va[j] < key;
            key = better ? va[j] : key;
            bz = better ? rx[j].z : bz; bw = better ? rx[j].w : bw; bi = better ? lane * c + j : bi;
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
        const int cnt = wave_count(cl >= 1) + wave_count(cl >= 2) + wave_count(cl >= 3);
        const unsigned uz = static_cast<unsigned>(lane_value(bz, L[0])), uw = static_cast<unsigned>(lane_value(bw, L[0]));
        const int ui = lane_value(bi, L[0]);
        int local, q1_, nr1, nq1;
        rec_unpack(uz, uw, local, q1_, nr1, nq1);
        const bool any = m[0] < dinf();
        if (lane == 0) { s_dec.v1 = m[0]; s_dec.cnt = cnt; s_dec.r1 = any ? ui * kCols + local : -1; s_dec.q1 = any ? q1_ : -1; s_dec.nr1 = any ? nr1 : -1; s_dec.nq1 = any ? nq1 : -1; }
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
        if (wave == 0) reduce_minimum_packed(std::integral_constant<int, kC4>{}, q0, qok);
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
    // A lane's four centroid elements are two PAIRS of neighbours, 2 lane + 128 r + {0, 1} (round 6): two 16-byte requests per centroid instead of four of
    // 8 bytes (4.99 -> 4.93 us per round at 43 200 points; the squared difference below is a tree sum in any case).  Rows of odd d start on 8-byte boundaries
    // (the unaligned access mode serves the 16-byte read) and their last element, which has no partner, is handled on its own in the merge phase.
    // STRAIGHT-LINE code on purpose: the same reads behind `if (d even) .. else ..`, or through a packed 8-byte-aligned pair type with a one-element
    // branch, compiled into rounds of 5.6 / 5.9 us (profiles/r06_round_requests_probe.txt).
    auto celem = [lane](const int j) { return 2 * lane + 128 * (j >> 1) + (j & 1); };
    auto cload = [&](const double *c, double (&x)[kCk]) {
#pragma unroll
        for (int r = 0; r < kCk / 2; ++r) {
            const int k = 2 * lane + 128 * r;
            const bool in = k + 1 < d;
            double2 q;
            if constexpr (kUncond) q = *reinterpret_cast<const double2 *>(c + (in ? k : 0));
            else q = in ? *reinterpret_cast<const double2 *>(c + k) : make_double2(0.0, 0.0);
            x[2 * r] = in ? q.x : 0.0; x[2 * r + 1] = in ? q.y : 0.0;
        }
    };
    const bool spec = FA_AHC_SPECULATE && R1 >= 0 && Q1 >= 0;
    const bool spec_lo = R1 < Q1;
    // without a pair to speculate on (a re-scan round) the unconditional form requests slot / node 0 and drops the values
    const int sp_a = !spec ? 0 : (spec_lo ? R1 : Q1), sp_b = !spec ? 0 : (spec_lo ? Q1 : R1), sp_na = !spec ? 0 : (spec_lo ? NR1 : NQ1), sp_nb = !spec ? 0 : (spec_lo ? NQ1 : NR1);
    double sp_ma = 0.0, sp_mb = 0.0, sp_da[CPT], sp_db[CPT], sp_xa[kCk], sp_xb[kCk];
#pragma unroll
    for (int j = 0; j < kCk; ++j) { sp_xa[j] = 0.0; sp_xb[j] = 0.0; }
#pragma unroll
    for (int j = 0; j < CPT; ++j) { sp_da[j] = 0.0; sp_db[j] = 0.0; }
    // (the youngest load of the round's first batch is consumed here: the load counter completes in order, so everything older has arrived and
    // nothing in the decision below has to wait on the counter — a wait there would also wait for the requests that follow)
    asm volatile("" :: "v"(nanflag), "v"(e2x[CPT - 1]), "v"(rs[CPT - 1].d1), "v"(nx[CPT - 1]));
    if (kUncond || spec)
    {
        // sizes and centroids first, the two matrix entries (a cold row each) last: loads complete in order, the centroid arithmetic runs under the entries'
        // latency.  (Entries first measured in round 6: 5.12 against 5.07 us.  Requesting the entries from every thread, round 4: dead columns then read cold
        // column copies nobody needs — 5.8 instead of 5.3 us per round at 43 200 points; here a column that wants nothing reads its ROW copy, next to its
        // neighbours' entries.)
        sp_ma = w.sizes[sp_na]; sp_mb = w.sizes[sp_nb];
        const double *ca = w.C + static_cast<size_t>(sp_na) * d, *cb = w.C + static_cast<size_t>(sp_nb) * d;
        cload(ca, sp_xa); cload(cb, sp_xb);
        __builtin_amdgcn_sched_barrier(0);
        if (kUncond || st.mode == FA_AHC_MODE_AUTO) {
            pair_entries<CPT, kEntryBranch>(w.M, Np, sp_a, sp_na, x0, nx, st.sym_limit, sp_a, sp_b, sp_da);
            pair_entries<CPT, kEntryBranch>(w.M, Np, sp_b, sp_nb, x0, nx, st.sym_limit, sp_a, sp_b, sp_db);
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

    if (D.op != OP_MERGE) {
        // a round that does something else drops the speculative operands: consumed here, or their requests stay "pending" for the compiler on this path and
        // the join in front of the block record waits for the whole in-order counter — on the MERGE path that is a wait for the round's STORES (round 6)
        asm volatile("" :: "v"(sp_ma), "v"(sp_mb));
#pragma unroll
        for (int j = 0; j < kCk; ++j) asm volatile("" :: "v"(sp_xa[j]), "v"(sp_xb[j]));
#pragma unroll
        for (int j = 0; j < CPT; ++j) asm volatile("" :: "v"(sp_da[j]), "v"(sp_db[j]));
    }
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
                pair_entries<CPT, kEntryBranch>(w.M, Np, a, na, x0, nx, st.sym_limit, a, b, da);
                pair_entries<CPT, kEntryBranch>(w.M, Np, b, nb, x0, nx, st.sym_limit, a, b, db);
            }
            // consumed INSIDE the (rare) branch: loads still pending at the join make the compiler wait for the whole in-order counter at the first use
            // behind it — on the common path that was a wait for the two cold matrix entries in front of the centroid arithmetic (round 6, from the listing)
            asm volatile("" :: "v"(ma), "v"(mb));
#pragma unroll
            for (int j = 0; j < CPT; ++j) asm volatile("" :: "v"(da[j]), "v"(db[j]));
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
            cload(ca, xa); cload(cb, xb);
#pragma unroll
            for (int j = 0; j < kCk; ++j) asm volatile("" :: "v"(xa[j]), "v"(xb[j]));   // (as above)
        }
        AHC_STAMP(9);
        const bool keeps_centroid = wave == 0 && (st.mode == FA_AHC_MODE_EXACT || blk == 0);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < kCk; ++j) {
            const int k = celem(j);
            if (keeps_centroid && k < d) {
                const double cc = __ddiv_rn(__dadd_rn(__dmul_rn(xa[j], ma), __dmul_rn(xb[j], mb)), den);
                if (st.mode == FA_AHC_MODE_EXACT) s_cvec[k] = cc;
                if (blk == 0) w.C[static_cast<size_t>(nnew) * d + k] = cc;
            }
            const double diff = xa[j] - xb[j];
            part += diff * diff;
        }
        // d > 256: the elements from 256 on, 64 at a time; odd d <= 256: the last element (it has no partner in the pair reads) through the same loop — no
        // conditional region of its own: a join behind one with memory operations costs a wait on the in-order counter in EVERY round (measured: 0.15 us)
        for (int k = ((d & 1) != 0 && d <= 64 * kCk ? d - 1 : 64 * kCk) + lane; k < d; k += 64) {
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
            pair_entries<CPT, kEntryBranch>(w.M, Np, S, pnode_[k], x0, nx, st.sym_limit, S, -1, ent);
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
template <bool BATCH, bool BIG, int CPT = 1, int KC = 4 / CPT>
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
    ahc_round_body<false, BIG, CPT, KC>(w_, blk_, ph);
}

// A problem of at most 256 points is ONE block: its rounds need no device-wide barrier at all, a workgroup barrier between them (with
// the release / acquire that makes the records and row states written by some threads visible to the others) is enough — all rounds
// of a replay in one launch, no kernel boundary, operands in the local caches (agent-scope fences around the barrier were measured
// 0.5 us per round slower and are not needed inside one workgroup).
template <int CPT>   // up to kBlk * CPT points
__global__ __launch_bounds__(kBlk) void ahc_rounds_single_block(const Ws w, const int rounds) {
    for (int r = 0; r < rounds; ++r) {
        ahc_round_body<false, false, CPT, 1>(w, 0, r & 3);   // one block: one record
        __syncthreads();   // workgroup-scope release / acquire: the waves of one workgroup share the CU's caches
    }
}


}  // namespace fa_ahc
