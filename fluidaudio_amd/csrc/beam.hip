// beam.hip — CTC prefix beam search with word-level ARPA language model rescoring, batched on gfx950.
//
// Replaces ctcBeamSearch (reference: Sources/FluidAudio/ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:118-241) and
// ARPALanguageModel (…/CTC/ARPALanguageModel.swift:16-104).  One workgroup per utterance walks the frames; inside a frame
// everything is data-parallel over the <= beamWidth x (tokenCandidates + 1) candidate hypotheses:
//
//   1. top-K tokens of the frame, (log-prob descending, index ascending): computed for ALL frames ahead of the walk by ctc_topk_kernel
//      (one wavefront per frame, the whole chip busy) — the walk reads K (token, log-prob) pairs per frame;
//   2. candidate totals.  The reference merges hypotheses through a dictionary keyed by the whole token prefix; here a prefix
//      is a node of a trie kept in HBM — an open-addressing hash table over (parent node, token), so that a prefix which is
//      pruned and later re-created gets the SAME node — and two observations replace the dictionary: an extension
//      (beam i, token v) can only collide with the ONE live beam whose (parent node, last token) is (node i, v), and never
//      with another extension.  Two 256-slot LDS hash maps (node -> beam, (parent, token) -> beam) answer both questions;
//   3. pruning to the beam width: the candidates are (total descending, candidate order ascending) keys in registers; every wavefront
//      bounds its own by a binary search over the key bits, wavefront 0 repeats that over the gathered ones; rank sort of the <= 128 survivors;
//   4. survivors become the new beams; new prefixes get trie nodes, and — when a language model is attached — their
//      running word (a polynomial hash of its bytes, built from per-token (multiplier, addend) pairs) is scored once
//      against the unigram / bigram hash tables in HBM, so a frame costs at most beamWidth table probes.
//
// Order-dependent details of the reference (it iterates Swift dictionaries) are resolved as in the CPU restatement
// (oracle.ctc_beam_search): candidates are ordered beam-major / token-minor, a merged hypothesis takes the earlier position,
// ties keep that order.  logAddExp is evaluated in double and rounded to float (within 1 ulp of the reference's Float libm).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "fa_common.h"
#include "text_util.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBeam = 128;
constexpr int kMaxTop = 64;
constexpr int kMapSlots = 256;
constexpr uint64_t kHashBase = 0x100000001b3ull;   // odd: multiplication by it is a bijection mod 2^64
constexpr float kUnkLogProb = -23.026f;            // ARPALanguageModel.unkLogProb (:33)

struct UniEntry { uint64_t h; int32_t len; float logp, backoff; int32_t used; };
struct BiEntry { uint64_t hp, hw; int32_t lp, lw; float logp; int32_t used; };
struct TokInfo { uint64_t mult, add; int32_t len, boundary; };   // stripped piece: h' = h * mult + add, len' = len + this len

struct LmView {
    const UniEntry *uni; const BiEntry *bi;
    uint32_t uni_mask, bi_mask;   // capacity - 1
};

__host__ __device__ inline uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__host__ __device__ inline bool uni_find(const LmView &lm, uint64_t h, int32_t len, float &logp, float &backoff) {
    if (!lm.uni) return false;
    for (uint32_t s = static_cast<uint32_t>(mix64(h + static_cast<uint64_t>(len))) & lm.uni_mask;; s = (s + 1) & lm.uni_mask) {
        const UniEntry e = lm.uni[s];
        if (!e.used) return false;
        if (e.h == h && e.len == len) { logp = e.logp; backoff = e.backoff; return true; }
    }
}

__host__ __device__ inline bool bi_find(const LmView &lm, uint64_t hp, int32_t lp, uint64_t hw, int32_t lw, float &logp) {
    if (!lm.bi) return false;
    for (uint32_t s = static_cast<uint32_t>(mix64(mix64(hp + static_cast<uint64_t>(lp)) ^ (hw + static_cast<uint64_t>(lw) * 0x9e3779b97f4a7c15ull))) & lm.bi_mask;;
         s = (s + 1) & lm.bi_mask) {
        const BiEntry e = lm.bi[s];
        if (!e.used) return false;
        if (e.hp == hp && e.hw == hw && e.lp == lp && e.lw == lw) { logp = e.logp; return true; }
    }
}

// ARPALanguageModel.score (:98-103); plen < 0 encodes prev == nil
__host__ __device__ inline float lm_score(const LmView &lm, uint64_t hw, int32_t lw, uint64_t hp, int32_t plen) {
    float logp, bo;
    if (plen >= 0 && bi_find(lm, hp, plen, hw, lw, logp)) return logp;
    float backoff = 0.0f;
    if (plen >= 0 && uni_find(lm, hp, plen, logp, bo)) backoff = bo;
    const float uni = uni_find(lm, hw, lw, logp, bo) ? logp : kUnkLogProb;
    return backoff + uni;
}

// The same score on the device with the three first probes (bigram, context unigram, word unigram) requested together: they are dependent
// HBM / L2 round trips on the serial frame walk, and the bigram usually misses.  Collisions continue with the sequential probes above.
__device__ inline float lm_score_dev(const LmView &lm, uint64_t hw, int32_t lw, uint64_t hp, int32_t plen) {
    if (!lm.uni) return lm_score(lm, hw, lw, hp, plen);
    const uint32_t sw = static_cast<uint32_t>(mix64(hw + static_cast<uint64_t>(lw))) & lm.uni_mask;
    const UniEntry ew = lm.uni[sw];
    if (plen < 0) {
        if (!ew.used) return kUnkLogProb;                       // backoff 0 + unknown word
        if (ew.h == hw && ew.len == lw) return 0.0f + ew.logp;
        return lm_score(lm, hw, lw, hp, plen);
    }
    const uint32_t sp = static_cast<uint32_t>(mix64(hp + static_cast<uint64_t>(plen))) & lm.uni_mask;
    const UniEntry ep = lm.uni[sp];
    bool bi_known = !lm.bi, bi_hit = false;
    float bi_logp = 0.0f;
    if (lm.bi) {
        const uint32_t sb = static_cast<uint32_t>(mix64(mix64(hp + static_cast<uint64_t>(plen)) ^ (hw + static_cast<uint64_t>(lw) * 0x9e3779b97f4a7c15ull))) & lm.bi_mask;
        const BiEntry eb = lm.bi[sb];
        if (!eb.used) bi_known = true;
        else if (eb.hp == hp && eb.hw == hw && eb.lp == plen && eb.lw == lw) { bi_known = true; bi_hit = true; bi_logp = eb.logp; }
    }
    const bool w_known = !ew.used || (ew.h == hw && ew.len == lw), p_known = !ep.used || (ep.h == hp && ep.len == plen);
    if (!(bi_known && w_known && p_known)) return lm_score(lm, hw, lw, hp, plen);   // a first slot held another key: probe on
    if (bi_hit) return bi_logp;
    const float backoff = ep.used ? ep.backoff : 0.0f;
    const float uni = ew.used ? ew.logp : kUnkLogProb;
    return backoff + uni;
}

inline void hash_bytes(const char *s, size_t n, uint64_t &h, uint64_t &mult) {
    h = 0; mult = 1;
    for (size_t i = 0; i < n; ++i) { h = h * kHashBase + static_cast<unsigned char>(s[i]); mult *= kHashBase; }
}

// monotone float -> uint map (larger float, larger uint); NaN sorts below everything
__device__ __forceinline__ uint32_t ord_f32(float f) {
    if (f != f) return 0u;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// max + log(exp(a - max) + exp(b - max)) evaluated in double and rounded to float.  exp(0) is exactly 1, so the sum is 1 + exp(-|a - b|)
// (one exp); and when |a - b| > 18.1 the correction log1p(exp(-d)) < 1.4e-8 is below half an ulp of any |max| >= 1, so the rounded
// result IS max — the same bits as the full evaluation, without the two libm calls (most (pb, pnb) pairs late in an utterance).
__device__ __forceinline__ float log_add_exp(float a, float b) {   // CtcDecoder.swift:279-284
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = fmaxf(a, b), d = fabsf(a - b);
    if (d > 18.1f && fabsf(m) >= 1.0f) return m;
    const double sum = 1.0 + exp(-static_cast<double>(d));
    return static_cast<float>(static_cast<double>(m) + log(sum));
}

struct TopEntry { int32_t tok; float lp; };   // tok: token id, bit 31 = the piece starts a word (TokInfo::boundary)

struct Beams {   // structure of arrays in LDS
    int32_t node[kMaxBeam], parent[kMaxBeam], last[kMaxBeam], wlen[kMaxBeam], plen[kMaxBeam];
    float pb[kMaxBeam], pnb[kMaxBeam], tot[kMaxBeam], lm[kMaxBeam], wscore[kMaxBeam];   // tot = logAddExp(pb, pnb) = totalAcoustic (:83), carried along
    uint64_t wh[kMaxBeam], ph[kMaxBeam];
};

struct BeamArgs {
    const float *logp; const int32_t *valid; const TokInfo *tok; LmView lm;
    unsigned long long *arena;   // [B][arena_stride] trie = hash table of (parent node << 32 | token), node id = slot; -1 = empty prefix
    int32_t *tokens, *lens; float *scores;
    int64_t row_stride, matrix_stride, arena_stride;
    int32_t frames, vocab, blank, beam_width, top_k, use_lm, first;
    float lm_weight, word_bonus;
    unsigned long long *prof;   // FA_BEAM_PROF (diagnostics): cycles per step of workgroup 0, thread 0
    const TopEntry *top;   // the pre-pass' table (ctc_topk_kernel): [workgroup][frame][top_k + 1]
};

struct Shared {
    Beams b[2];
    float stay_pb[kMaxBeam], stay_pnb[kMaxBeam], stay_tot[kMaxBeam];      // slot 0 of every beam: its blank / repeat parts and their logAddExp
    TopEntry top_e[kMaxTop];                                               // the frame's top tokens (padded behind the last one with NaN log-probs)
    int32_t top_len[kMaxTop];
    unsigned long long top_mult[kMaxTop], top_add[kMaxTop];                // word-hash step of a top token (TokInfo)
    unsigned long long cancel[kMaxBeam];                                   // bit r: extension r of the beam was merged into a live beam's slot 0
    unsigned long long wave_sel[kThreads / 64][kMaxBeam];                  // level 1 of the selection: each wavefront's best keys
    unsigned long long sel_key[kMaxBeam];
    int32_t rank_hi[kMaxBeam], wave_cnt[kThreads / 64];
    int32_t map_node[kMapSlots], map_node_val[kMapSlots];
    int32_t map_pl_parent[kMapSlots], map_pl_tok[kMapSlots], map_pl_val[kMapSlots];
    int32_t sel_count, n_beams;
    float blank_lp;
};

// OR over the wavefront of a 32-bit word (DPP row shifts + row broadcasts; lane 63 holds the result)
__device__ __forceinline__ unsigned wave_or(unsigned v) {
#define FA_BEAM_OR(CTRL, MASK) v |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, MASK, 0xf, false));
    FA_BEAM_OR(0x111, 0xf) FA_BEAM_OR(0x112, 0xf) FA_BEAM_OR(0x114, 0xf) FA_BEAM_OR(0x118, 0xf) FA_BEAM_OR(0x142, 0xa) FA_BEAM_OR(0x143, 0xc)
#undef FA_BEAM_OR
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// ascending sort of s.sel_key[0 .. 128) (real keys first, then the ~0ull padding).  The real keys are distinct (their low words are
// distinct orders), so the rank of a key = the number of smaller keys: 128 broadcast LDS reads per thread and ONE barrier, instead of the
// 28 barrier-separated stages of a bitonic network.
__device__ void sort_selected(Shared &s) {
    const int tid = threadIdx.x, me = tid & (kMaxBeam - 1), half = tid >> 7;   // two threads per key: each counts over one half of the keys
    static_assert(kThreads == 2 * kMaxBeam, "two threads per selected key");
    const unsigned long long mine = s.sel_key[me];
    int part = 0;
    if (mine != ~0ull) {
        const unsigned long long *src = s.sel_key + half * (kMaxBeam / 2);
#pragma unroll 16
        for (int j = 0; j < kMaxBeam / 2; ++j) part += src[j] < mine ? 1 : 0;
    }
    if (half) s.rank_hi[me] = part;
    __syncthreads();
    if (!half && mine != ~0ull) s.sel_key[part + s.rank_hi[me]] = mine;
    __syncthreads();
}

// inclusive prefix sum over the wavefront (DPP: Hillis-Steele inside the 16-lane rows, then the row totals)
__device__ __forceinline__ int wave_incl_scan(int v) {
#define FA_BEAM_ADD(CTRL, MASK) v += __builtin_amdgcn_update_dpp(0, v, CTRL, MASK, 0xf, false);
    FA_BEAM_ADD(0x111, 0xf) FA_BEAM_ADD(0x112, 0xf) FA_BEAM_ADD(0x114, 0xf) FA_BEAM_ADD(0x118, 0xf) FA_BEAM_ADD(0x142, 0xa) FA_BEAM_ADD(0x143, 0xc)
#undef FA_BEAM_ADD
    return v;
}

// ------------------------------------------------------------------------------------------------ top tokens of every frame (pre-pass)
// Which tokens a frame offers (CtcDecoder.swift:141-144: the tokenCandidates best by log-probability, ties by index, blank aside) does not
// depend on the beams, so it leaves the serial frame walk: ctc_topk_kernel computes it for every (utterance, frame) row of a launch at once —
// one wavefront per row, rows spread over the whole chip — and ctc_beam_kernel reads K (token, log-prob) pairs per frame, requested one
// frame ahead.  (Rounds 1-3 selected inside the walk: 11 900 of 57 000 cycles per frame step, profiles/r02_beam_probe.json.)
//
// One row: lane l holds the order keys u = ~ord_f32(x) (smaller = better; NaN = 0xffffffff sorts last) of the tokens l, l + 64, ...  A binary
// search over the bits of u, from the first bit in which the row's keys differ, keeps T with count(u < T) < K; it stops as soon as a tested
// bound H has K <= count(u < H) <= 64 — those keys are gathered and ranked, the first K kept — or when all 32 bits are fixed: then T is the
// K-th smallest key and ties at T enter in index order.  No barrier, no LDS atomics: DPP reductions, ballots and one 512-byte LDS slab per wave.

struct TopArgs {
    const float *logp; const int32_t *valid; const TokInfo *tok;
    TopEntry *top;                             // [utterance of the launch][frame][top_k + 1]: the K best tokens, then (lp) the blank's log-probability
    int64_t row_stride, matrix_stride, rows;
    int32_t frames, vocab, blank, top_k, first, use_lm;
};

__device__ __forceinline__ int wave_sum(int v) {   // total over the wavefront, in every lane's scalar copy
    return __builtin_amdgcn_readlane(wave_incl_scan(v), 63);
}
__device__ __forceinline__ unsigned wave_and(unsigned v) { return ~wave_or(~v); }

constexpr int kTopRegs = 17;   // keys per lane held in registers: rows of up to 1 088 tokens (Parakeet CTC: 1 025); longer rows are re-read (L2)

template <int NREG>   // NREG > 0: the row's keys live in registers; 0: every pass reads the row again
__global__ __launch_bounds__(kThreads) void ctc_topk_kernel(const TopArgs a) {
    __shared__ unsigned long long slab[kThreads / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rowi = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + wave;
    if (rowi >= a.rows) return;
    const int ul = static_cast<int>(rowi / a.frames), t = static_cast<int>(rowi - static_cast<int64_t>(ul) * a.frames), u = a.first + ul;
    if (a.valid) { const int v = a.valid[u]; if (t >= v) return; }
    const float *row = a.logp + static_cast<int64_t>(u) * a.matrix_stride + static_cast<int64_t>(t) * a.row_stride;
    const int V = a.vocab, K = a.top_k, blank = a.blank;
    const bool has_blank = blank >= 0 && blank < V;
    if (lane == 0) a.top[rowi * (K + 1) + K] = TopEntry{0, has_blank ? row[blank] : -INFINITY};
    const int npresent = V - (has_blank ? 1 : 0), ntop = npresent < K ? npresent : K;
    if (ntop <= 0) return;
    const int nj = (V + 63) >> 6;
    auto present = [&](const int j) { const int i = lane + 64 * j; return i < V && i != blank; };
    auto load_key = [&](const int j) -> unsigned { return present(j) ? ~ord_f32(row[lane + 64 * j]) : 0xffffffffu; };
    unsigned reg[NREG > 0 ? NREG : 1];
    if (NREG > 0) {
#pragma unroll
        for (int j = 0; j < NREG; ++j) reg[j] = load_key(j);
    }
    // f(j, key) over this lane's keys; absent positions carry 0xffffffff (never below any bound; `present` tells them from NaN)
#define FA_TOP_EACH(BODY)                                                                                  \
    if (NREG > 0) { _Pragma("unroll") for (int j = 0; j < NREG; ++j) { const unsigned key = reg[j]; BODY } } \
    else { for (int j = 0; j < nj; ++j) { const unsigned key = load_key(j); BODY } }
    unsigned bound = 0xffffffffu, tie = 0;   // selection: key < bound, plus (ties) the first `quota` present keys == tie in index order
    int quota = 0;
    if (npresent > 64) {
        unsigned kand = 0xffffffffu, kor = 0;
        FA_TOP_EACH(if (present(j)) { kand &= key; kor |= key; })
        kand = wave_and(kand); kor = wave_or(kor);
        const unsigned differ = kand ^ kor;
        int b = differ ? 31 - __clz(static_cast<int>(differ)) : -1;          // highest bit in which two present keys differ
        unsigned T = b >= 31 ? 0u : (b < 0 ? kor : (kor >> (b + 1)) << (b + 1));   // every present key is >= T: count(key < T) = 0
        int below = 0;
        bool found = false;
        for (; b >= 0 && !found; --b) {
            const unsigned cand = T | (1u << b);
            int c = 0;
            FA_TOP_EACH(c += key < cand ? 1 : 0;)
            c = wave_sum(c);
            if (c < K) { T = cand; below = c; }
            else if (c <= 64) { bound = cand; found = true; }
        }
        if (!found) { bound = T; tie = T; quota = K - below; }                 // T = the K-th smallest key; absent keys never tie (`present`)
    }
    // gather the selected keys (<= 64) into the wave's slab as (key, index) words, then rank them
    int cnt = 0, tcnt = 0;
    FA_TOP_EACH({
        const bool pr = present(j);
        const bool is_tie = quota > 0 && pr && key == tie;
        const unsigned long long tm = __ballot(is_tie);
        const int trank = tcnt + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(tm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(tm), 0));
        tcnt += __popcll(tm);
        const bool sel = pr && (npresent <= 64 || key < bound || (is_tie && trank < quota));
        const unsigned long long sm = __ballot(sel);
        const int pos = cnt + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(sm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(sm), 0));
        cnt += __popcll(sm);
        if (sel && pos < 64) slab[wave][pos] = (static_cast<unsigned long long>(key) << 32) | static_cast<unsigned>(lane + 64 * j);
    })
#undef FA_TOP_EACH
    cnt = cnt < 64 ? cnt : 64;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the slab is written by other lanes of this wavefront
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long mine = lane < cnt ? slab[wave][lane] : ~0ull;
    int rank = 0;
    for (int m = 0; m < cnt; ++m) rank += slab[wave][m] < mine ? 1 : 0;      // distinct words: the rank is a permutation
    if (lane < cnt && rank < ntop) {
        const int v = static_cast<int>(mine & 0xffffffffu);
        TopEntry e;
        e.tok = v | (a.use_lm && a.tok[v].boundary ? static_cast<int32_t>(0x80000000u) : 0);
        e.lp = row[v];
        a.top[rowi * (K + 1) + rank] = e;
    }
}

// ---------------------------------------------------------------------------------------------- selection of the beam width's best candidates
// The candidates of a frame never touch LDS: the two threads of a beam compute its extension totals straight into 64-bit keys in registers
// (value descending, candidate order ascending; ~0 = absent), and the W smallest keys are found in two wave-level steps without histograms:
//   level 1: every wavefront bounds ITS keys — a binary search over the key bits from the first bit in which they differ keeps T with
//            count(key < T) < W and stops at the first tested bound with W <= count <= 128 (DPP sums, no barrier, no LDS atomics) — and
//            gathers the <= 128 keys below the bound by ballot prefix sums;
//   level 2: wavefront 0 does the same over the <= 4 x 128 gathered keys (8 per lane) and leaves <= 128 keys in s.sel_key;
//   then the rank sort above.  The W best of the whole frame are among the per-wave W best, so the result is the one of a full selection.
// (Rounds 2-3: candidate totals through LDS arrays, two 8-bit radix selections over LDS histograms, ~20 barriers: 22 300 cycles per frame.)
// A key is two 32-bit words (kh: value, descending; kl: candidate order); absent = both ~0 (a present key's value word is never ~0: that would
// be a NaN total, and those are dropped).
struct KeyBound { unsigned h, l; };   // the keys below (h, l) are selected

template <int NK>
__device__ __forceinline__ KeyBound wave_bound(const unsigned (&kh)[NK], const unsigned (&kl)[NK], const int k) {
    int pc = 0;                                                          // per-lane counts (v_cmp + v_addc), one DPP sum per step: counting
#pragma unroll                                                           // through ballots serialises on the VALU -> SALU latency (870 cycles per step)
    for (int j = 0; j < NK; ++j) pc += kh[j] != ~0u ? 1 : 0;
    pc = wave_sum(pc);
    if (pc <= kMaxBeam) return KeyBound{~0u, ~0u};                      // everything present fits the gather buffer
    unsigned ah = ~0u, al = ~0u, oh = 0, ol = 0;
#pragma unroll
    for (int j = 0; j < NK; ++j) { const bool pr = kh[j] != ~0u; ah &= kh[j]; al &= pr ? kl[j] : ~0u; oh |= pr ? kh[j] : 0u; ol |= pr ? kl[j] : 0u; }
    ah = wave_and(ah); oh = wave_or(oh);
    const unsigned dh = ah ^ oh;
    unsigned Th = oh;
    if (dh) {                                                            // binary search over the value word, from the first bit in which the keys differ
        int b = 31 - __clz(static_cast<int>(dh));
        Th = b >= 31 ? 0u : (oh >> (b + 1)) << (b + 1);
        for (; b >= 0; --b) {
            const unsigned cand = Th | (1u << b);
            int c = 0;
#pragma unroll
            for (int j = 0; j < NK; ++j) c += kh[j] < cand ? 1 : 0;
            c = wave_sum(c);
            if (c < k) Th = cand;
            else if (c <= kMaxBeam) return KeyBound{cand, 0u};
        }
    }
    // rare: the k-th place falls inside a run of equal values (Th) — or more than 128 keys share it: the order word decides
    al = wave_and(al); ol = wave_or(ol);
    const unsigned dl = dh ? ~0u : (al ^ ol);                            // != 0: more than 128 distinct keys
    int b = 31 - __clz(static_cast<int>(dl));
    unsigned Tl = b >= 31 ? 0u : (ol >> (b + 1)) << (b + 1);
    for (; b >= 0; --b) {
        const unsigned cand = Tl | (1u << b);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NK; ++j) c += (kh[j] < Th) | ((kh[j] == Th) & (kl[j] < cand)) ? 1 : 0;
        c = wave_sum(c);
        if (c < k) Tl = cand;
        else if (c <= kMaxBeam) return KeyBound{Th, cand};
    }
    return KeyBound{Th, Tl + 1u};   // all bits fixed: the keys are distinct, (Th, Tl) is the k-th smallest and exactly k keys lie below (Th, Tl + 1)
}

// gathers this wavefront's keys below `bound` into dst[0 ..), lane after lane (their order does not matter: they are sorted later): every lane
// counts its own, one DPP prefix sum gives its first position
template <int NK>
__device__ __forceinline__ int wave_gather(const unsigned (&kh)[NK], const unsigned (&kl)[NK], const KeyBound bound, unsigned long long *dst) {
    bool sel[NK];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < NK; ++j) { sel[j] = (kh[j] < bound.h) | ((kh[j] == bound.h) & (kl[j] < bound.l)); mine += sel[j] ? 1 : 0; }   // bound (~0, ~0): every present key
    const int incl = wave_incl_scan(mine);
    int pos = incl - mine;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        if (sel[j] && pos < kMaxBeam) dst[pos] = (static_cast<unsigned long long>(kh[j]) << 32) | kl[j];
        pos += sel[j] ? 1 : 0;
    }
    const int cnt = __builtin_amdgcn_readlane(incl, 63);
    return cnt < kMaxBeam ? cnt : kMaxBeam;
}

// value word of a key: ~ord_f32(x) for a non-NaN x (larger total, smaller word)
__device__ __forceinline__ unsigned desc_word(const float x) {
    const unsigned u = __float_as_uint(x);
    return u ^ (static_cast<unsigned>(~static_cast<int>(u) >> 31) & 0x7fffffffu);
}

// slot of a 64-bit key (a, b) in the 256-slot LDS maps: 32-bit multiplies only (the maps compare whole keys, the hash only spreads them)
__device__ __forceinline__ uint32_t slot_of(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9e3779b1u ^ b * 0x85ebca77u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    return h & (kMapSlots - 1);
}

constexpr int kOrdShift = 7;   // candidate order = beam << 7 | slot (slot 0 = the beam itself, 1 + r = its extension by top token r): beam-major, token-minor

// MAXE: extension keys per thread = ceil(top tokens / 2)
template <int MAXE, bool PROF>   // PROF: FA_BEAM_PROF cycle stamps (their 16 accumulators cost 32 registers)
__global__ __launch_bounds__(kThreads) void ctc_beam_kernel(const BeamArgs a) {
    __shared__ Shared s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = a.first + blockIdx.x;
    int T = a.frames;
    if (a.valid) { const int v = a.valid[u]; T = v < 0 ? 0 : (v < T ? v : T); }
    unsigned long long *arena = a.arena + static_cast<int64_t>(blockIdx.x) * a.arena_stride;
    const uint32_t arena_mask = static_cast<uint32_t>(a.arena_stride - 1);
    const int W = a.beam_width, K = a.top_k, V = a.vocab;

    if (tid == 0) {
        Beams &b = s.b[0];
        b.node[0] = -1; b.parent[0] = -3; b.last[0] = -1; b.wlen[0] = 0; b.plen[0] = -1;
        b.pb[0] = 0.0f; b.pnb[0] = -INFINITY; b.tot[0] = 0.0f; b.lm[0] = 0.0f; b.wscore[0] = 0.0f; b.wh[0] = 0; b.ph[0] = 0;
        s.n_beams = 1;
    }
    __syncthreads();

    unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = clock64();
#define BEAM_STAMP(i) do { if (PROF) { const unsigned long long t_now = clock64(); t_acc[i] += t_now - t_prev; t_prev = t_now; } } while (0)
    int cur = 0;
    // the frame's top tokens, best first (sorted { frame[$0] > frame[$1] }, stable: ties by index; CtcDecoder.swift:141-144): computed for all
    // frames by ctc_topk_kernel; thread r carries entry r of the frame, requested one frame ahead so that its HBM latency hides behind the frame
    const bool has_blank = a.blank >= 0 && a.blank < V;
    const int ntop = min(K, V - (has_blank ? 1 : 0));
    // (entry K of a frame is the blank's log-probability, carried by thread 64: a per-lane request like the others — as a scalar load it would
    // share the LDS' counter, and the frame's first LDS wait would sit out its HBM latency)
    const TopEntry *top = a.top + static_cast<int64_t>(blockIdx.x) * a.frames * (K + 1);
    const bool carries = tid < ntop || tid == kMaxTop;
    const int my_entry = tid == kMaxTop ? K : tid;
    TopEntry e_next = TopEntry{0, 0.0f};
    if (T > 0 && carries) e_next = top[my_entry];
    const int bi = tid >> 1, h = tid & 1;                                // this thread's beam and which half of its extensions
    // The trie node of a NEW beam (thread = its rank) is found / inserted by an atomic in HBM: a dependent round trip of ~1.5 us on the serial walk.
    // Nothing needs the node before the next frame's maps, so the atomic is issued LAST in step 5 and its answer is collected in the next frame
    // behind the extension keys (which need no nodes): the round trip travels under a barrier, the top-token table and the key arithmetic.
    bool pend = false;
    uint32_t pend_q = 0;
    unsigned long long pend_seen = 0, pend_nk = 0;
    auto resolve_node = [&](Beams &bb) {
        if (pend) {
            while (pend_seen != ~0ull && pend_seen != pend_nk) { pend_q = (pend_q + 1) & arena_mask; pend_seen = atomicCAS(&arena[pend_q], ~0ull, pend_nk); }
            bb.node[tid] = static_cast<int>(pend_q);
            pend = false;
        }
    };
    for (int t = 0; t < T; ++t) {
        Beams &b = s.b[cur];
        Beams &nb = s.b[cur ^ 1];
        const int n = s.n_beams;
        // ---- 1. top-token table of the frame into LDS; empty maps ----
        TokInfo ti_mine = TokInfo{1, 0, 0, 0};
        if (tid < kMaxTop) {
            TopEntry e = e_next;
            if (tid >= ntop) { e.tok = 0x7ffffffe; e.lp = __uint_as_float(0x7fc00000u); }   // padding: equals no beam's last token; NaN total = absent candidate
            s.top_e[tid] = e;
            if (a.use_lm && tid < ntop) ti_mine = a.tok[e.tok & 0x7fffffff];   // the word-hash step of the token: used by step 5, most of a frame of latency cover
        }
        if (tid == kMaxTop) s.blank_lp = e_next.lp;
        if (t + 1 < T && carries) e_next = top[static_cast<int64_t>(t + 1) * (K + 1) + my_entry];
        s.map_node[tid] = -2;
        s.map_pl_parent[tid] = -4;
        if (tid < kMaxBeam) s.cancel[tid] = 0ull;
        __syncthreads();
        BEAM_STAMP(0);
        // ---- 2. extension keys (threads 2 i, 2 i + 1: beam i); maps over the live beams (thread i: beam i) ----
        unsigned kh[MAXE + 1], kl[MAXE + 1];                            // this thread's candidate keys: extensions 2 q + h of its beam, then (h = 0) the beam itself
        int r_last = -1;                                                 // the top token equal to the beam's last token (at most one)
        int my_last = -1;
        float my_tot = 0.0f, my_lm = 0.0f;
        if (bi < n) {
            my_last = b.last[bi];
            my_tot = b.tot[bi]; my_lm = b.lm[bi];
            const float my_pb = b.pb[bi];
            const float lm_word = b.wlen[bi] > 0 ? my_lm + b.wscore[bi] : my_lm;   // a word is completed by a boundary piece (:185-189)
            TopEntry e[MAXE];
#pragma unroll
            for (int q = 0; q < MAXE; ++q) e[q] = s.top_e[2 * q + h];    // all requested before the first is used; padded entries give NaN totals
#pragma unroll
            for (int q = 0; q < MAXE; ++q) {
                const int r = 2 * q + h;
                const bool repeat = (e[q].tok & 0x7fffffff) == my_last;
                r_last = repeat ? r : r_last;
                const float pnb = (repeat ? my_pb : my_tot) + e[q].lp;                           // (:196-214)
                const float total = pnb + (e[q].tok < 0 ? lm_word : my_lm);
                const bool ok = total == total;
                kh[q] = ok ? desc_word(total) : ~0u;
                kl[q] = ok ? static_cast<unsigned>((bi << kOrdShift) | (1 + r)) : ~0u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXE; ++q) { kh[q] = ~0u; kl[q] = ~0u; }
        }
        kh[MAXE] = ~0u; kl[MAXE] = ~0u;
        resolve_node(b);                                                 // this thread's beam got its trie node (requested at the end of the previous frame)
        if (tid < n) {
            const int node = b.node[tid], par = b.parent[tid], last = b.last[tid];
            for (uint32_t q = slot_of(static_cast<uint32_t>(node), 0x51u);; q = (q + 1) & (kMapSlots - 1))
                if (atomicCAS(&s.map_node[q], -2, node) == -2) { s.map_node_val[q] = tid; break; }
            for (uint32_t q = slot_of(static_cast<uint32_t>(par), static_cast<uint32_t>(last));; q = (q + 1) & (kMapSlots - 1))
                if (atomicCAS(&s.map_pl_parent[q], -4, par) == -4) { s.map_pl_tok[q] = last; s.map_pl_val[q] = tid; break; }
        }
        {   // the pair's two halves agree on r_last (quad_perm 1,0,3,2: the neighbour lane)
            const int other = __builtin_amdgcn_update_dpp(-1, r_last, 0xB1, 0xf, 0xf, false);
            r_last = r_last > other ? r_last : other;
        }
        __syncthreads();
        BEAM_STAMP(2);
        // ---- 3. the beam itself (slot 0): blank + repeated token, merged with the extension of the live beam one token shorter ----
        auto find_node = [&](const int node) {
            for (uint32_t q = slot_of(static_cast<uint32_t>(node), 0x51u);; q = (q + 1) & (kMapSlots - 1)) {
                const int k = s.map_node[q];
                if (k == -2) return -1;
                if (k == node) return s.map_node_val[q];
            }
        };
        auto find_child = [&](const int parent, const int tok) {   // live beam whose prefix is (parent prefix) + tok
            for (uint32_t q = slot_of(static_cast<uint32_t>(parent), static_cast<uint32_t>(tok));; q = (q + 1) & (kMapSlots - 1)) {
                const int k = s.map_pl_parent[q];
                if (k == -4) return -1;
                if (k == parent && s.map_pl_tok[q] == tok) return s.map_pl_val[q];
            }
        };
        if (bi < n && h == 0) {
            float pnb = -INFINITY;
            uint32_t ord = static_cast<uint32_t>(bi << kOrdShift);
            if (r_last >= 0) {
                const float lp = s.top_e[r_last].lp;
                pnb = b.pnb[bi] + lp;                                                // same prefix, repeated token (:190-194)
                const int par = b.parent[bi];
                const int p = par != -3 ? find_node(par) : -1;                       // the beam one token shorter extends into this one
                if (p >= 0) {
                    const float from_parent = (b.last[p] == my_last ? b.pb[p] : b.tot[p]) + lp;
                    pnb = log_add_exp(pnb, from_parent);
                    const uint32_t merged = static_cast<uint32_t>((p << kOrdShift) | (1 + r_last));
                    ord = merged < ord ? merged : ord;                               // a merged hypothesis takes the earlier position
                    atomicOr(&s.cancel[p], 1ull << r_last);                          // that extension is this candidate now
                }
            }
            const float pb = my_tot + s.blank_lp;                                    // blank extension (:172-176)
            const float stay = log_add_exp(pb, pnb);
            s.stay_pb[bi] = pb; s.stay_pnb[bi] = pnb; s.stay_tot[bi] = stay;
            const float total = stay + my_lm;
            if (total == total) { kh[MAXE] = desc_word(total); kl[MAXE] = ord; }
        }
        __syncthreads();
        BEAM_STAMP(3);
        // ---- 4. prune: W best totals, earlier candidate first on ties ----
        if (bi < n) {
            const unsigned long long gone = s.cancel[bi] >> h;
            const unsigned g0 = static_cast<unsigned>(gone), g1 = static_cast<unsigned>(gone >> 32);   // 32-bit tests (64-bit compares run at quarter rate)
#pragma unroll
            for (int q = 0; q < MAXE; ++q) {
                const bool c = (((2 * q < 32 ? g0 >> ((2 * q) & 31) : g1 >> ((2 * q - 32) & 31)) & 1u) != 0u);
                kh[q] = c ? ~0u : kh[q]; kl[q] = c ? ~0u : kl[q];
            }
        }
        BEAM_STAMP(8);
        {
            const KeyBound bound = wave_bound<MAXE + 1>(kh, kl, W);
            BEAM_STAMP(9);
            const int cnt = wave_gather<MAXE + 1>(kh, kl, bound, s.wave_sel[wave]);
            if (lane == 0) s.wave_cnt[wave] = cnt;
        }
        BEAM_STAMP(10);
        if (a.use_lm && tid < ntop) { s.top_mult[tid] = ti_mine.mult; s.top_add[tid] = ti_mine.add; s.top_len[tid] = ti_mine.len; }   // requested in step 1
        __syncthreads();
        BEAM_STAMP(4);
        if (wave == 0) {
            constexpr int N2 = 2 * (kThreads / 64);
            unsigned h2[N2], l2[N2];
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                const int w = j >> 1, idx = lane + 64 * (j & 1);
                const unsigned long long kk = idx < s.wave_cnt[w] ? s.wave_sel[w][idx] : ~0ull;
                h2[j] = static_cast<unsigned>(kk >> 32); l2[j] = static_cast<unsigned>(kk);
            }
            const KeyBound bound = wave_bound<N2>(h2, l2, W);
            const int cnt = wave_gather<N2>(h2, l2, bound, s.sel_key);
            for (int i = cnt + lane; i < kMaxBeam; i += 64) s.sel_key[i] = ~0ull;
            if (lane == 0) s.sel_count = cnt;
        }
        __syncthreads();
        BEAM_STAMP(5);
        sort_selected(s);
        const int nsel = min(s.sel_count, W);
        BEAM_STAMP(6);
        // ---- 5. survivors -> new beams (rank = sorted position) ----
        if (tid < nsel) {
            const unsigned ord = static_cast<unsigned>(s.sel_key[tid] & 0xffffffffu);
            // recover the candidate: order values of extensions are their own; a beam's slot 0 may carry the order of the extension merged into it
            int i = static_cast<int>(ord >> kOrdShift), slot = static_cast<int>(ord & ((1u << kOrdShift) - 1u));
            if (slot != 0) {
                const int j = find_child(b.node[i], s.top_e[slot - 1].tok & 0x7fffffff);
                if (j >= 0) { i = j; slot = 0; }                                         // it was the merged position of beam j
            }
            if (slot == 0) {
                nb.node[tid] = b.node[i]; nb.parent[tid] = b.parent[i]; nb.last[tid] = b.last[i];
                nb.pb[tid] = s.stay_pb[i]; nb.pnb[tid] = s.stay_pnb[i]; nb.tot[tid] = s.stay_tot[i]; nb.lm[tid] = b.lm[i];
                nb.wlen[tid] = b.wlen[i]; nb.plen[tid] = b.plen[i]; nb.wh[tid] = b.wh[i]; nb.ph[tid] = b.ph[i]; nb.wscore[tid] = b.wscore[i];
            } else {
                const int r = slot - 1, v = s.top_e[r].tok & 0x7fffffff;
                // canonical trie node of (prefix of beam i) + v: find or insert; at most frames x beam_width nodes, table twice that
                const unsigned long long nk = (static_cast<unsigned long long>(static_cast<uint32_t>(b.node[i])) << 32) | static_cast<uint32_t>(v);
                const float pnb = (v == b.last[i] ? b.pb[i] : b.tot[i]) + s.top_e[r].lp;
                nb.parent[tid] = b.node[i]; nb.last[tid] = v;
                nb.pb[tid] = -INFINITY; nb.pnb[tid] = pnb; nb.tot[tid] = pnb;           // logAddExp(-inf, x) = x
                float lm = b.lm[i], ws = 0.0f;
                uint64_t wh = b.wh[i], ph = b.ph[i];
                int wlen = b.wlen[i], plen = b.plen[i];
                if (a.use_lm) {
                    if (s.top_e[r].tok < 0) {                                            // a boundary piece (:183-194)
                        if (wlen > 0) { lm = lm + b.wscore[i]; ph = wh; plen = wlen; }
                        wh = s.top_add[r]; wlen = s.top_len[r];
                    } else { wh = wh * s.top_mult[r] + s.top_add[r]; wlen += s.top_len[r]; }   // wordPieces.append(piece) (:195-197)
                    if (wlen > 0) ws = a.lm_weight * lm_score_dev(a.lm, wh, wlen, ph, plen) + a.word_bonus;
                }
                nb.lm[tid] = lm; nb.wscore[tid] = ws; nb.wh[tid] = wh; nb.ph[tid] = ph; nb.wlen[tid] = wlen; nb.plen[tid] = plen;
                pend = true; pend_nk = nk; pend_q = static_cast<uint32_t>(mix64(nk)) & arena_mask;   // behind the model's probes: memory answers in order
                pend_seen = atomicCAS(&arena[pend_q], ~0ull, nk);
            }
        }
        if (tid == 0) s.n_beams = nsel;
        cur ^= 1;
        __syncthreads();
        BEAM_STAMP(7);
    }
    if (PROF && a.prof && blockIdx.x == 0 && tid == 0 && a.first == 0) for (int i = 0; i < 16; ++i) a.prof[i] = t_acc[i];
#undef BEAM_STAMP
    resolve_node(s.b[cur]);
    __syncthreads();

    // ---- finalize: trailing partial word (:222-229), first maximum in rank order, read the prefix back from the trie ----
    if (tid == 0) {
        const Beams &b = s.b[cur];
        int best = -1;
        float best_total = 0.0f;
        for (int i = 0; i < s.n_beams; ++i) {
            float lm = b.lm[i];
            if (a.use_lm && b.wlen[i] > 0) lm = lm + b.wscore[i];
            const float total = b.tot[i] + lm;
            if (best < 0 || total > best_total) { best = i; best_total = total; }
        }
        int32_t *out = a.tokens + static_cast<int64_t>(u) * a.frames;
        int len = 0;
        if (best >= 0 && T > 0) {
            for (int node = b.node[best]; node >= 0; node = static_cast<int>(arena[node] >> 32)) ++len;
            int pos = len;
            for (int node = b.node[best]; node >= 0; node = static_cast<int>(arena[node] >> 32)) out[--pos] = static_cast<int32_t>(arena[node] & 0xffffffffu);
        }
        a.lens[u] = len;
        if (a.scores) a.scores[u] = T > 0 && best >= 0 ? best_total : 0.0f;
    }
}

inline uint32_t pow2_at_least(size_t n) { uint32_t c = 16; while (c < n) c <<= 1; return c; }

bool parse_float_full(const std::string &s, float &out) { return fa_text::parse_float(s, out); }   // Swift's Float(String)

}  // namespace

struct fa_arpa_lm {
    fa_ctx *ctx = nullptr;
    std::vector<UniEntry> uni;
    std::vector<BiEntry> bi;
    int64_t n_uni = 0, n_bi_ctx = 0, n_bi = 0;
    void *d_uni = nullptr, *d_bi = nullptr;
    int dev_id = -1;   // device holding d_uni / d_bi
    LmView host_view() const { return LmView{uni.empty() ? nullptr : uni.data(), bi.empty() ? nullptr : bi.data(), static_cast<uint32_t>(uni.size() - 1), static_cast<uint32_t>(bi.size() - 1)}; }
    LmView dev_view() const { return LmView{static_cast<const UniEntry *>(d_uni), static_cast<const BiEntry *>(d_bi), static_cast<uint32_t>(uni.size() - 1), static_cast<uint32_t>(bi.size() - 1)}; }
};

struct fa_ctc_vocab {
    fa_ctx *ctx = nullptr;
    int32_t vocab_size = 0;
    void *d_tok = nullptr;
};

extern "C" {

fa_status fa_arpa_parse(fa_ctx *ctx, const char *text, int64_t len, fa_arpa_lm **out) {
    if (!out || (!text && len > 0) || len < 0) return FA_INVALID_ARGUMENT;   // ctx may be NULL: parsing and scoring are host code
    *out = nullptr;
    try {
        const float log10_to_nat = static_cast<float>(std::log(10.0));           // ARPALanguageModel.log10ToNat (:30)
        struct U { std::string w; float p, b; };
        struct B2 { std::string c, w; float p; };
        std::unordered_map<std::string, size_t> uidx;
        std::unordered_map<std::string, size_t> bidx;
        std::vector<U> us;
        std::vector<B2> bs;
        std::unordered_map<std::string, int> contexts;
        std::string section;
        for (int64_t pos = 0; pos <= len;) {
            int64_t e = pos;
            while (e < len && text[e] != '\n') ++e;                                              // the reader cuts at the byte \n only (:126)
            const char *la = text + pos, *lb = text + e;
            pos = e + 1;
            fa_text::trim(la, lb, fa_text::ws_or_nl_len);                                        // trimmingCharacters(in: .whitespacesAndNewlines) (:131)
            const std::string line(la, lb);
            if (line.empty() || line.rfind("\\data\\", 0) == 0) continue;                       // :54
            if (line == "\\end\\") break;                                                       // :55
            if (line[0] == '\\') { section = line; continue; }                                  // :56-59
            if (line.rfind("ngram ", 0) == 0) continue;                                         // :61
            std::vector<std::string> parts;
            for (size_t i = 0;;) { const size_t j = line.find('\t', i); parts.emplace_back(line.substr(i, j == std::string::npos ? j : j - i)); if (j == std::string::npos) break; i = j + 1; }
            float l10;
            if (!parse_float_full(parts[0], l10)) continue;                                     // malformed line skipped (:64-67)
            const float prob = l10 * log10_to_nat;
            auto backoff = [&](size_t i) { float v; return parts.size() > i ? (parse_float_full(parts[i], v) ? v : 0.0f) * log10_to_nat : 0.0f; };
            if (section == "\\1-grams:" && parts.size() >= 2) {
                const float bo = backoff(2);
                auto it = uidx.find(parts[1]);
                if (it == uidx.end()) { uidx[parts[1]] = us.size(); us.push_back({parts[1], prob, bo}); } else { us[it->second].p = prob; us[it->second].b = bo; }
            } else if (section == "\\2-grams:" && parts.size() >= 3) {
                const std::string key = parts[1] + '\t' + parts[2];
                auto it = bidx.find(key);
                if (it == bidx.end()) { bidx[key] = bs.size(); bs.push_back({parts[1], parts[2], prob}); } else bs[it->second].p = prob;
                contexts[parts[1]] = 1;
            }
        }
        fa_arpa_lm *lm = new fa_arpa_lm();
        lm->ctx = ctx; lm->n_uni = static_cast<int64_t>(us.size()); lm->n_bi = static_cast<int64_t>(bs.size()); lm->n_bi_ctx = static_cast<int64_t>(contexts.size());
        lm->uni.assign(pow2_at_least(2 * us.size() + 1), UniEntry{0, 0, 0.f, 0.f, 0});
        lm->bi.assign(pow2_at_least(2 * bs.size() + 1), BiEntry{0, 0, 0, 0, 0.f, 0});
        const uint32_t um = static_cast<uint32_t>(lm->uni.size() - 1), bm = static_cast<uint32_t>(lm->bi.size() - 1);
        for (const U &x : us) {
            uint64_t h, m; hash_bytes(x.w.data(), x.w.size(), h, m);
            const int32_t l = static_cast<int32_t>(x.w.size());
            uint32_t s = static_cast<uint32_t>(mix64(h + static_cast<uint64_t>(l))) & um;
            while (lm->uni[s].used) s = (s + 1) & um;
            lm->uni[s] = UniEntry{h, l, x.p, x.b, 1};
        }
        for (const B2 &x : bs) {
            uint64_t hp, hw, m; hash_bytes(x.c.data(), x.c.size(), hp, m); hash_bytes(x.w.data(), x.w.size(), hw, m);
            const int32_t lp = static_cast<int32_t>(x.c.size()), lw = static_cast<int32_t>(x.w.size());
            uint32_t s = static_cast<uint32_t>(mix64(mix64(hp + static_cast<uint64_t>(lp)) ^ (hw + static_cast<uint64_t>(lw) * 0x9e3779b97f4a7c15ull))) & bm;
            while (lm->bi[s].used) s = (s + 1) & bm;
            lm->bi[s] = BiEntry{hp, hw, lp, lw, x.p, 1};
        }
        *out = lm;
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "arpa: host allocation failed");
    } catch (...) {
        return fa::set_error(ctx, FA_UNKNOWN_ERROR, "arpa: unexpected failure");
    }
}

void fa_arpa_destroy(fa_arpa_lm *lm) {
    if (!lm) return;
    if (lm->d_uni || lm->d_bi) { fa::DeviceGuard guard(lm->dev_id); (void)hipFree(lm->d_uni); (void)hipFree(lm->d_bi); }
    delete lm;
}

int64_t fa_arpa_unigram_count(const fa_arpa_lm *lm) { return lm ? lm->n_uni : 0; }
int64_t fa_arpa_bigram_context_count(const fa_arpa_lm *lm) { return lm ? lm->n_bi_ctx : 0; }

fa_status fa_arpa_score(const fa_arpa_lm *lm, const char *word, const char *prev, float *out) {
    if (!lm || !word || !out) return FA_INVALID_ARGUMENT;
    uint64_t hw, hp = 0, m;
    hash_bytes(word, strlen(word), hw, m);
    if (prev) hash_bytes(prev, strlen(prev), hp, m);
    *out = lm_score(lm->host_view(), hw, static_cast<int32_t>(strlen(word)), hp, prev ? static_cast<int32_t>(strlen(prev)) : -1);
    return FA_SUCCESS;
}

fa_status fa_ctc_vocab_create(fa_ctx *ctx, const int32_t *ids, const char *const *pieces, int32_t n, int32_t vocab_size, fa_ctc_vocab **out) {
    if (!ctx || !out || n < 0 || vocab_size < 1 || (n > 0 && (!ids || !pieces))) return FA_INVALID_ARGUMENT;
    *out = nullptr;
    std::vector<TokInfo> tok(vocab_size, TokInfo{1, 0, 0, 0});                        // missing id: vocabulary[v] ?? "" (:181)
    static const char kBoundary[] = "\xe2\x96\x81";                                   // U+2581, ASRConstants.sentencePieceWordBoundary
    for (int32_t i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= vocab_size || !pieces[i]) continue;
        const char *p = pieces[i];
        size_t len = strlen(p);
        TokInfo t{1, 0, 0, 0};
        if (len >= 3 && memcmp(p, kBoundary, 3) == 0) { t.boundary = 1; p += 3; len -= 3; }   // hasPrefix + dropFirst (:184,:192)
        hash_bytes(p, len, t.add, t.mult);
        t.len = static_cast<int32_t>(len);
        tok[ids[i]] = t;
    }
    fa::DeviceGuard guard(ctx->device);
    fa_ctc_vocab *v = new (std::nothrow) fa_ctc_vocab();
    if (!v) return FA_ALLOCATION_FAILURE;
    v->ctx = ctx; v->vocab_size = vocab_size;
    hipError_t e = hipMalloc(&v->d_tok, sizeof(TokInfo) * vocab_size);
    if (e == hipSuccess) e = hipMemcpy(v->d_tok, tok.data(), sizeof(TokInfo) * vocab_size, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(v->d_tok); delete v; return fa::hip_status(ctx, e, "ctc vocab upload"); }
    *out = v;
    return FA_SUCCESS;
}

void fa_ctc_vocab_destroy(fa_ctc_vocab *v) {
    if (!v) return;
    if (v->d_tok) { fa::DeviceGuard guard(v->ctx->device); (void)hipFree(v->d_tok); }
    delete v;
}

fa_status fa_ctc_beam_search_batch_dev(fa_ctx *ctx, const float *d_log_probs, int32_t batch, int32_t frames, int32_t vocab, int64_t row_stride,
                                       int64_t matrix_stride, const int32_t *d_valid_frames, const fa_ctc_vocab *vocabulary, fa_arpa_lm *lm,
                                       int32_t beam_width, float lm_weight, float word_bonus, int32_t blank_id, int32_t token_candidates,
                                       int32_t *d_tokens, int32_t *d_lens, float *d_scores) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (batch == 0) return FA_SUCCESS;
    if (batch < 0 || frames < 0 || vocab < 1 || !d_tokens || !d_lens || (frames > 0 && !d_log_probs))
        return fa::set_error(ctx, FA_INVALID_ARGUMENT, "beam search: bad arguments");
    if (beam_width < 1 || beam_width > kMaxBeam || token_candidates < 0 || token_candidates > kMaxTop)
        return fa::set_error(ctx, FA_INVALID_ARGUMENT, "beam search: beam width 1..%d, token candidates 0..%d", kMaxBeam, kMaxTop);
    if (lm && (!vocabulary || vocabulary->vocab_size < vocab)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "beam search: the language model needs a vocabulary covering all tokens");
    if (row_stride < vocab) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "beam search: row stride < vocab");
    fa::DeviceGuard guard(ctx->device);
    if (lm && (!lm->d_uni || lm->dev_id != ctx->device)) {   // first use on this device: upload the tables (all or nothing)
        if (lm->d_uni || lm->d_bi) {   // tables of another device: release them there
            fa::DeviceGuard other(lm->dev_id);
            (void)hipDeviceSynchronize();
            (void)hipFree(lm->d_uni); (void)hipFree(lm->d_bi);
            lm->d_uni = nullptr; lm->d_bi = nullptr;
        }
        void *du = nullptr, *db = nullptr;
        hipError_t e = hipMalloc(&du, sizeof(UniEntry) * lm->uni.size());
        if (e == hipSuccess) e = hipMalloc(&db, sizeof(BiEntry) * lm->bi.size());
        if (e == hipSuccess) e = hipMemcpy(du, lm->uni.data(), sizeof(UniEntry) * lm->uni.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(db, lm->bi.data(), sizeof(BiEntry) * lm->bi.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(du); (void)hipFree(db); return fa::hip_status(ctx, e, "arpa table upload"); }
        lm->d_uni = du; lm->d_bi = db;
        lm->ctx = ctx; lm->dev_id = ctx->device;
    }
    BeamArgs a{};
    a.logp = d_log_probs; a.valid = d_valid_frames; a.tok = vocabulary ? static_cast<const TokInfo *>(vocabulary->d_tok) : nullptr;
    if (lm) a.lm = lm->dev_view();
    a.tokens = d_tokens; a.lens = d_lens; a.scores = d_scores;
    a.row_stride = row_stride; a.matrix_stride = matrix_stride;
    a.frames = frames; a.vocab = vocab; a.blank = blank_id; a.beam_width = beam_width; a.top_k = token_candidates;
    a.use_lm = lm != nullptr; a.lm_weight = lm_weight; a.word_bonus = word_bonus;
    a.arena_stride = pow2_at_least(static_cast<size_t>(2) * frames * beam_width + 2);
    // trie tables: one per utterance in flight, at most ~2 GiB at a time
    const int64_t per = a.arena_stride * static_cast<int64_t>(sizeof(unsigned long long));
    const int chunk = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(batch, (int64_t(2) << 30) / std::max<int64_t>(per, 1))));
    fa::DevBuf d_arena;
    FA_HIP_TRY(ctx, d_arena.alloc(ctx, static_cast<size_t>(per) * chunk));   // the context's buffer cache: a second call pays no hipMalloc
    a.arena = d_arena.as<unsigned long long>();
    fa::DevBuf d_prof;
    if (fa::sw(fa::Sw::BEAM_PROF)) { FA_HIP_TRY(ctx, d_prof.alloc(128)); FA_HIP_TRY(ctx, hipMemsetAsync(d_prof.p, 0, 128, ctx->stream)); a.prof = d_prof.as<unsigned long long>(); }
    // the pre-pass' table of one launch: K (token, log-prob) pairs + the blank's log-prob per frame
    fa::DevBuf d_top;
    const size_t rows_max = static_cast<size_t>(chunk) * std::max(frames, 1);
    FA_HIP_TRY(ctx, d_top.alloc(ctx, sizeof(TopEntry) * rows_max * (token_candidates + 1)));
    a.top = d_top.as<TopEntry>();
    TopArgs ta{};
    ta.logp = d_log_probs; ta.valid = d_valid_frames; ta.tok = a.tok; ta.top = d_top.as<TopEntry>();
    ta.row_stride = row_stride; ta.matrix_stride = matrix_stride; ta.frames = frames; ta.vocab = vocab; ta.blank = blank_id;
    ta.top_k = token_candidates; ta.use_lm = a.use_lm;
    if (ctx->timing) FA_HIP_TRY(ctx, hipEventRecord(ctx->tim_ev[0], ctx->stream));   // device work of the call: behind the allocations
    for (int first = 0; first < batch; first += chunk) {
        const int now = std::min(chunk, batch - first);
        a.first = first;
        FA_HIP_TRY(ctx, hipMemsetAsync(d_arena.p, 0xff, static_cast<size_t>(per) * now, ctx->stream));
        if (frames > 0) {
            ta.first = first; ta.rows = static_cast<int64_t>(now) * frames;
            const unsigned blocks = static_cast<unsigned>((ta.rows + kThreads / 64 - 1) / (kThreads / 64));
            if (vocab <= 64 * kTopRegs) hipLaunchKernelGGL(ctc_topk_kernel<kTopRegs>, dim3(blocks), dim3(kThreads), 0, ctx->stream, ta);
            else hipLaunchKernelGGL(ctc_topk_kernel<0>, dim3(blocks), dim3(kThreads), 0, ctx->stream, ta);
            FA_HIP_TRY(ctx, hipGetLastError());
        }
        const int ntop = std::min(token_candidates, vocab - (blank_id >= 0 && blank_id < vocab ? 1 : 0));   // extension keys per thread = ceil(ntop / 2)
        if (a.prof && ntop <= 16) hipLaunchKernelGGL((ctc_beam_kernel<8, true>), dim3(now), dim3(kThreads), 0, ctx->stream, a);
        else if (a.prof && ntop <= 40) hipLaunchKernelGGL((ctc_beam_kernel<20, true>), dim3(now), dim3(kThreads), 0, ctx->stream, a);
        else if (a.prof) hipLaunchKernelGGL((ctc_beam_kernel<32, true>), dim3(now), dim3(kThreads), 0, ctx->stream, a);
        else if (ntop <= 16) hipLaunchKernelGGL((ctc_beam_kernel<8, false>), dim3(now), dim3(kThreads), 0, ctx->stream, a);
        else if (ntop <= 40) hipLaunchKernelGGL((ctc_beam_kernel<20, false>), dim3(now), dim3(kThreads), 0, ctx->stream, a);
        else hipLaunchKernelGGL((ctc_beam_kernel<32, false>), dim3(now), dim3(kThreads), 0, ctx->stream, a);
        FA_HIP_TRY(ctx, hipGetLastError());
    }
    if (ctx->timing) FA_HIP_TRY(ctx, hipEventRecord(ctx->tim_ev[1], ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the arena and the tables go back to the context's cache on return
    if (ctx->timing) { float ms = -1.0f; FA_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->tim_ev[0], ctx->tim_ev[1])); ctx->last_device_ms = ms; }
    if (a.prof) {
        unsigned long long h[16];
        FA_HIP_TRY(ctx, hipMemcpy(h, d_prof.p, 128, hipMemcpyDeviceToHost));
        const double f = frames > 0 ? frames : 1;
        fprintf(stderr, "beam profile (cycles per frame, workgroup 0): top-token table + empty maps %.0f | maps + extension keys %.0f | slot 0 of the beams %.0f | "
                        "selection level 1 %.0f (cancelled extensions %.0f, bound %.0f, gather %.0f, barrier %.0f) | level 2 %.0f | rank sort %.0f | new beams %.0f\n",
                h[0] / f, h[2] / f, h[3] / f, (h[4] + h[8] + h[9] + h[10]) / f, h[8] / f, h[9] / f, h[10] / f, h[4] / f, h[5] / f, h[6] / f, h[7] / f);
    }
    return FA_SUCCESS;
}

// What a call of these shapes launches (bench.py prints it next to its timing): out = { trie slots per utterance (arena_stride), utterances
// per launch (the ~2 GiB arena cap), launches, extension keys per thread of the ctc_beam_kernel<MAXE, false> instance (8 / 20 / 32) }.
fa_status fa_ctc_beam_plan(int32_t batch, int32_t frames, int32_t vocab, int32_t beam_width, int32_t blank_id, int32_t token_candidates, int64_t out[4]) {
    if (!out || batch < 0 || frames < 0 || vocab < 1 || beam_width < 1 || beam_width > kMaxBeam || token_candidates < 0 || token_candidates > kMaxTop) return FA_INVALID_ARGUMENT;
    const int64_t stride = static_cast<int64_t>(pow2_at_least(static_cast<size_t>(2) * frames * beam_width + 2));
    const int64_t per = stride * static_cast<int64_t>(sizeof(unsigned long long));
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(batch, (int64_t(2) << 30) / std::max<int64_t>(per, 1)));
    const int ntop = std::min(token_candidates, vocab - (blank_id >= 0 && blank_id < vocab ? 1 : 0));
    out[0] = stride; out[1] = chunk; out[2] = batch > 0 ? (batch + chunk - 1) / chunk : 0; out[3] = ntop <= 16 ? 8 : (ntop <= 40 ? 20 : 32);
    return FA_SUCCESS;
}

fa_status fa_ctc_beam_search_batch(fa_ctx *ctx, const float *log_probs, int32_t batch, int32_t frames, int32_t vocab, const int32_t *valid_frames,
                                   const fa_ctc_vocab *vocabulary, fa_arpa_lm *lm, int32_t beam_width, float lm_weight, float word_bonus,
                                   int32_t blank_id, int32_t token_candidates, int32_t *tokens, int32_t *lens, float *scores) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (batch == 0) return FA_SUCCESS;
    if (batch < 0 || frames < 0 || vocab < 1 || !tokens || !lens || (frames > 0 && !log_probs)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "beam search: bad arguments");
    fa::DeviceGuard guard(ctx->device);
    const size_t n = static_cast<size_t>(batch) * frames * vocab;
    fa::DevBuf d_lp, d_valid, d_tok, d_len, d_sc;
    FA_HIP_TRY(ctx, d_lp.alloc(sizeof(float) * n));
    FA_HIP_TRY(ctx, d_tok.alloc(sizeof(int32_t) * static_cast<size_t>(batch) * std::max(frames, 1)));
    FA_HIP_TRY(ctx, d_len.alloc(sizeof(int32_t) * batch));
    FA_HIP_TRY(ctx, d_sc.alloc(sizeof(float) * batch));
    if (n) FA_HIP_TRY(ctx, hipMemcpyAsync(d_lp.p, log_probs, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    if (valid_frames) {
        FA_HIP_TRY(ctx, d_valid.alloc(sizeof(int32_t) * batch));
        FA_HIP_TRY(ctx, hipMemcpyAsync(d_valid.p, valid_frames, sizeof(int32_t) * batch, hipMemcpyHostToDevice, ctx->stream));
    }
    FA_TRY(fa_ctc_beam_search_batch_dev(ctx, d_lp.as<float>(), batch, frames, vocab, vocab, static_cast<int64_t>(frames) * vocab,
                                        valid_frames ? d_valid.as<int32_t>() : nullptr, vocabulary, lm, beam_width, lm_weight, word_bonus, blank_id,
                                        token_candidates, d_tok.as<int32_t>(), d_len.as<int32_t>(), d_sc.as<float>()));
    if (frames > 0) FA_HIP_TRY(ctx, hipMemcpyAsync(tokens, d_tok.p, sizeof(int32_t) * static_cast<size_t>(batch) * frames, hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipMemcpyAsync(lens, d_len.p, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, ctx->stream));
    if (scores) FA_HIP_TRY(ctx, hipMemcpyAsync(scores, d_sc.p, sizeof(float) * batch, hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FA_SUCCESS;
}

}  // extern "C"
