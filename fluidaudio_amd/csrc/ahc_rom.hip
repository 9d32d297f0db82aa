// ahc_rom.hip — the reference's selection order with the Lance-Williams matrix as the filter of its scans (ahc_ws.h: the map).
#include "ahc_ws.h"

using namespace fa_ahc;

namespace {
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
    int32_t nan_seen, kind, scanned, sa, sb, created, last_tie, tie_nonzero; // what the next scan launch computes: the row of node `scanned` (slot sa); last_tie: the
                                                                           // row count when an exact tie was last seen (equal sums among a scan's candidates, a height equal to the
                                                                           // previous row's, a row that went to exact sums), tie_nonzero: one of them while the merges were above height 0
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
    bool row_tie = st.kind == ROM_EXACT;       // (a row that needed exact sums of every workgroup: more than kRomCap near-ties)
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
            if (lane == 0) { st.kind = ROM_EXACT; st.exact_scans = st.exact_scans + 1; st.last_tie = st.merges; if (st.dab != 0.0) st.tie_nonzero = 1; w.dev[ph ^ 1] = st; }
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
                if (bm < dinf() && (__popcll(__builtin_amdgcn_ballot_w64(mine && sv == bm)) >= 2 || (bm == best && bi != best_id))) row_tie = true;
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
    const double prev_height = st.dab;
    const int prev_merges = st.merges;
    rom_prepare(st, sel, w.slot_of, w.sizes);
    // tie bookkeeping for the host's hand-over decision (rom_run_device): this scan's candidates, and a merge at exactly the previous merge's height
    // (tie_nonzero: a tie while the merges are at a height > 0.  Duplicated points tie — among a scan's candidates at ANY distance, two copies being equally far
    // from everything — only until the copies have merged, and all of those merges are at height 0; a tie seen while the dendrogram is above 0 is of the other kind)
    if (st.kind == ROM_NEW && st.merges > prev_merges && prev_merges >= 1 && st.dab == prev_height) row_tie = true;
    if (row_tie) { st.last_tie = st.merges; if (st.dab != 0.0) st.tie_nonzero = 1; }
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

}  // namespace

namespace fa_ahc {
// Workspace of the matrix-filtered run: the selection's arrays, then what the start-up kernels of the filter-based rounds expect (points / centroids,
// transpose, norms, the two state records their maxima go to), the matrix last.
struct RomLayout { size_t prof, dev, flags, part, part2, node, slot, sizes, key, ent, pos, ngh, next, prev, pa, pb, hs, z, state, norms, c, xt, m, total; Layout core; };
// What this run shares with the filter-based rounds sits WHERE THEY KEEP IT (make_layout_core with one slot per thread: state, flags, node, sizes, dendrogram,
// norms, centroids, transpose, matrix), its own arrays behind their last byte: the rounds can take the problem over in place (prob_adopt).
RomLayout rom_layout(size_t N, size_t Np, size_t d) {
    RomLayout L{};
    L.core = make_layout_core(N, Np, d, Np / kBlk);
    L.state = L.core.state; L.flags = L.core.flags; L.node = L.core.node; L.sizes = L.core.sizes; L.z = L.core.z; L.norms = L.core.norms;
    L.c = L.core.c; L.xt = L.core.xt; L.m = L.core.m;
    size_t o = L.core.total;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~static_cast<size_t>(255); return at; };
    const size_t nblk = Np / kBlk;
    L.dev = take(sizeof(RomDev) * 2); L.prof = take(8 * 16);
    L.part = take(sizeof(RomPart) * nblk); L.part2 = take(8 * nblk);
    L.slot = take(4 * 2 * N); L.key = take(8 * N); L.ent = take(sizeof(fa_ro::Ent) * N); L.pos = take(4 * 2 * N);
    L.ngh = take(4 * 2 * N); L.next = take(4 * (2 * N + 1)); L.prev = take(4 * (2 * N + 1));
    L.pa = take(8 * N); L.pb = take(8 * N); L.hs = take(8 * N);
    L.total = o;
    return L;
}

size_t rom_total_bytes(size_t N, size_t Np, size_t d) { return rom_layout(N, Np, d).total; }

// The whole problem in the reference's selection order with the matrix as the filter of its scans.  `declined` (no error recorded): the matrix cannot be
// had, or the Gram-form start-up met a non-finite entry (infinite coordinates: the sums of the matrix-free run decide what they mean) — the caller runs
// the matrix-free form instead.
// Handing the problem back (round 6).  An input whose exact ties are DUPLICATES (repeated embeddings, digital silence) ties at distance 0 only: once the copies
// have merged, the rest of the problem has a unique closest pair at every step, and every algorithm that merges the closest pair — the reference's and the
// filter-based rounds alike — produces the same rows.  So AUTO's tie route (may_hand_over) watches the ties go by (RomDev::last_tie / tie_nonzero) and, after
// kRomQuiet rows without one, none of them at a distance > 0 so far, and enough rows left to pay for it, lets the rounds adopt the state in place
// (prob_adopt: both runs keep node ids, sizes, centroids, matrix and dendrogram in the same arrays) at 5 us per row instead of 12.8.  The rounds detect an
// exact tie at the minimum themselves (need_exact): should one still come, the whole problem is run again in reference order WITHOUT handing over —
// slower than never having tried, never different from the reference.  Inputs that tie at distances > 0 (quantised embeddings) stay in reference order.
constexpr int kRomQuiet = 1024, kRomMinRest = 4096;

fa_status rom_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host, bool &declined, bool may_hand_over,
                         bool &tie_after_hand_over, bool matrix_ready = false) {
    tie_after_hand_over = false;
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
    FA_HIP_TRY(ctx, hipMemsetAsync(base + L.dev, 0, L.part - L.dev, st));              // RomDev, the profile counters
    FA_HIP_TRY(ctx, hipMemsetAsync(base + L.flags, 0, 16, st));
    // matrix_ready (AUTO's attempt halted before its first merge, in these very arrays): points, transpose, norms, Gram-form matrix and the start-up's maxima
    // (the cold part of state[0], which the rounds never write) are what this start-up would compute again — 10.6 of its 14 ms at 43 200 x 256
    if (!matrix_ready) {
        FA_HIP_TRY(ctx, hipMemsetAsync(base + L.state, 0, sizeof(AhcState) * 2, st));  // the two state records (the start-up's maxima start at 0)
        FA_HIP_TRY(ctx, hipMemcpyAsync(w.C, d_data, sizeof(double) * N * d, hipMemcpyDeviceToDevice, st));
    }
    ro_launch_init(st, rw, std::max(Np, 2 * N));
    if (!matrix_ready) startup_transpose(st, d_data, w.XT, w.N, w.Np, w.d);
    const bool direct_start = fa::sw_on(fa::Sw::AHC_ROM_DIRECT_START);   // the start-up of the matrix-free run (all N^2 / 2 exact sums) for A/B
    if (direct_start) ro_launch_lower_minima_direct(st, rw);
    if (!matrix_ready) FA_TRY(startup_gram(ctx, st, gw, d_norms));
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
        if (fa::sw(fa::Sw::AHC_DEBUG) && (it % 8 == 0 || hd.done))
            fprintf(stderr, "ahc (reference order): replay %lld rows %d last tie at %d nonzero %d scans %lld exact %lld\n", it, hd.merges, hd.last_tie, hd.tie_nonzero, hd.scans, hd.exact_scans);
        if (may_hand_over && !hd.done && !hd.nan_seen && !hd.tie_nonzero && hd.merges >= 1 && hd.merges - hd.last_tie >= kRomQuiet &&
            static_cast<long long>(N) - 1 - hd.merges >= kRomMinRest && hd.kind != ROM_EXACT) {
            // the state of a replay boundary sits in dev[0] (a replay is an even number of launch pairs).  A merge that is decided but not applied yet (ROM_NEW
            // pending: its row, node id, size and centroid are the next scan's work) is applied by that scan; its column copies stay unwritten, which is
            // what prob_adopt's sym_limit says.
            if (hd.kind == ROM_NEW) hipLaunchKernelGGL(rom_scan, dim3(w.nblk + 1), dim3(kBlk), lds, st, w, 0);
            Prob p;
            p.N = N; p.d = d; p.Np = Np; p.cpt = 1; p.d_data = d_data; p.d_Z = d_Z; p.mode = FA_AHC_MODE_AUTO; p.z_on_host = z_on_host;
            p.L = L.core;
            prob_bind(p, base);
            FA_TRY(prob_adopt(ctx, p, hd.merges, hd.eps, w.pair_a, w.pair_b));
            FA_TRY(prob_run_rounds(ctx, p));
            if (p.needs_ro) { tie_after_hand_over = true; return FA_SUCCESS; }   // an exact tie after all: the caller runs the problem again, in reference order to the end
            FA_HIP_TRY(ctx, hipEventRecord(ev[2], st));
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (fa::sw(fa::Sw::AHC_DEBUG))
                fprintf(stderr, "ahc (reference order, matrix filter): N %zu handed over to the rounds after %d rows (%lld scans), last tie at row %d; %lld rounds\n", N, hd.merges,
                        hd.scans, hd.last_tie, static_cast<long long>(p.h.rounds));
            if (stats) {
                float t01 = 0, t12 = 0;
                (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
                (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
                stats->merges = p.h.step; stats->rounds += hd.scans + hd.exact_scans + p.h.rounds; if (!stats->reference_order) stats->reference_order = 1;
                stats->rescans += hd.exact_scans + p.h.rescans; stats->windows += p.h.windows; stats->handed_over_at = hd.merges;
                stats->init_ms += t01; stats->merge_ms += t12; stats->total_ms += t01 + t12;
            }
            return FA_SUCCESS;
        }
    }
    if (hd.nan_seen == 1) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: NaN distance");
    if (!hd.done || hd.nan_seen || hd.merges != static_cast<int32_t>(N) - 1) return fa::set_error(ctx, FA_RUNTIME_ERROR, "ahc: reference-order run stopped at row %d", hd.merges);
    ro_launch_finish(st, rw);
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
fa_status ro_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, fa_ahc_stats *stats, bool z_on_host, bool may_hand_over, bool matrix_ready) {
    if (!fa::sw_on(fa::Sw::AHC_RO_NO_MATRIX)) {
        bool declined = false, tie_again = false;
        const fa_ahc_stats before = stats ? *stats : fa_ahc_stats{};
        fa_status st = rom_run_device(ctx, d_data, N, d, d_Z, stats, z_on_host, declined, may_hand_over && !fa::sw_on(fa::Sw::AHC_RO_NO_HANDOVER), tie_again, matrix_ready);
        if (!declined && st == FA_SUCCESS && tie_again) {   // the rounds met an exact tie after the hand-over: once more, in reference order to the last row
            if (stats) *stats = before;
            st = rom_run_device(ctx, d_data, N, d, d_Z, stats, z_on_host, declined, false, tie_again);
            if (stats && !declined) stats->handed_over_at = -1;
        }
        if (!declined) return st;
    }
    return ro_run_device_mf(ctx, d_data, N, d, d_Z, stats, z_on_host);
}

}  // namespace fa_ahc
