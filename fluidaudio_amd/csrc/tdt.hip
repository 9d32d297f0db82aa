// tdt.hip — Parakeet-TDT greedy decoding: frame navigation + the token/duration control loop (gfx950).
//
// Restates the control flow of TdtDecoderV3.decodeWithTimings (reference:
// Sources/FluidAudio/ASR/Parakeet/SlidingWindow/TDT/Decoder/TdtDecoderV3.swift:103-607) and its helpers
// (TdtFrameNavigation.swift:20-105, TdtDurationMapping.swift:17-31, TdtConfig.swift:13-26).
//
// In the reference every step of that loop crosses into CoreML: the decoder LSTM (one call per emitted token) and the
// joint network, which already returns the ARG-MAXED (token id, token probability, duration bin)
// (TdtModelInference.swift:107-138).  Those networks are not part of the reference tree, so what can be reproduced is
// the integer control flow.  The device entry therefore consumes the joint's decisions as tables indexed by
// (u = decoder steps taken so far in this chunk, t = encoder frame) — exactly the values the reference would see along
// its greedy path — and replays the loop for a whole batch of chunks, one thread per chunk (the loop is a serial walk of
// <= T + maxTokens table look-ups; the batch is the parallel axis).  PARITY of token outputs against the real models is
// UNPINNED (models absent); the navigation helpers are pinned by TdtRefactoredComponentsTests.swift:12-195.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "fa_common.h"

namespace {

constexpr int kStandardOverlapFrames = 25;  // ASRConstants.standardOverlapFrames (Shared/ASRConstants.swift:49)

struct TdtArgs {
    const int32_t *tok, *bin;  // [B][U][T]
    const float *prob;         // [B][U][T]
    const int32_t *enc_len, *audio_frames, *t0, *is_last, *global_offset, *emit_after;  // [B]; emit_after < 0: emit all
    int32_t *out_tok, *out_time, *out_dur;  // [B][max_out]
    float *out_conf;                        // [B][max_out]
    int32_t *out_count, *final_time, *final_u, *status;  // [B]
    int32_t B, U, T, max_out;
    fa_tdt_config cfg;
};

__host__ __device__ inline float clamp_probability(const float v) {  // TdtDurationMapping.swift:28-31
    if (!(v - v == 0.0f)) return 0.0f;  // NaN / +-inf
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

// The control loop for chunk b.  `raw(u, frame, tok, bin)` supplies one joint decision — the token and the duration bin, which decide the next
// cell —, `prob_of_last()` the probability of the token of the MOST RECENT decision, asked for only when that token is emitted (the reference's
// joint returns it with every decision, TdtModelInference.swift:107-138, but only an emitted token's confidence is ever used, :409-463): three of
// four decisions are blanks, and the soft-max denominator is most of a decision's arithmetic.  `writer` is true for the thread that stores the
// outputs (the table walk runs one thread per chunk; the logits walk runs a wavefront per chunk through the same, wave-uniform, control flow).
template <class Raw, class Prob>
__device__ __forceinline__ void tdt_walk(const TdtArgs &a, const int b, const bool writer, Raw &&raw, Prob &&prob_of_last) {
    auto uni = [](const int x) { return x; };
    const fa_tdt_config &c = a.cfg;
    int32_t *otok = a.out_tok + static_cast<int64_t>(b) * a.max_out, *otime = a.out_time + static_cast<int64_t>(b) * a.max_out;
    int32_t *odur = a.out_dur + static_cast<int64_t>(b) * a.max_out;
    float *oconf = a.out_conf + static_cast<int64_t>(b) * a.max_out;
    int count = 0, st = FA_SUCCESS, u = 0;
    const int enc_len = uni(a.enc_len[b]);
    const int goff = uni(a.global_offset ? a.global_offset[b] : 0);
    const int emit_after = uni(a.emit_after ? a.emit_after[b] : -1);
    int t = uni(a.t0 ? a.t0[b] : 0);
    if (writer) a.final_time[b] = INT32_MIN;  // "timeJump not updated" (early returns, :110-112,:150-152)
    auto finish = [&]() { if (writer) { a.out_count[b] = count; a.final_u[b] = u; a.status[b] = st; } };
    if (enc_len <= 1) { finish(); return; }                        // :110-112
    const int Teff = min(enc_len, uni(a.audio_frames ? a.audio_frames[b] : enc_len));  // TdtFrameNavigation.swift:59-78
    if (t >= Teff) { finish(); return; }                           // :150-152
    const int last = Teff - 1;
    int safe = min(t, last);
    bool active = t < Teff;
    int last_emit_t = -1, n_at_t = 0, processed = 0, tok = c.blank_id, dur = 0;
    float score = 0.0f;
    auto joint = [&](const int frame) -> bool {  // one joint decision; false on a table / duration-bin error
        if (u >= a.U || frame < 0 || frame >= a.T) { st = FA_OUTPUT_TOO_SMALL; return false; }
        int bi = 0;
        raw(u, frame, tok, bi);
        tok = uni(tok); bi = uni(bi);
        if (bi < 0 || bi >= c.n_duration_bins) { st = FA_RUNTIME_ERROR; return false; }  // mapDurationBin throws (:17-22)
        // (a dynamic index into the by-value config goes through a private copy — a per-lane load whose result counts as divergent and drags the whole
        // walk into vector registers; eight selects on the uniform bin keep it scalar)
        int dsel = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) dsel = bi == q ? c.duration_bins[q] : dsel;
        dur = uni(dsel);
        return true;
    };
    auto emit = [&](const int ts) {
        if (emit_after >= 0 && ts < emit_after) return;  // shouldEmitToken (:600-606)
        score = clamp_probability(prob_of_last());
        if (count < a.max_out) { if (writer) { otok[count] = tok; otime[count] = ts; odur[count] = dur; oconf[count] = score; } }
        else st = FA_OUTPUT_TOO_SMALL;
        ++count;
    };
    while (active) {  // :230-467
        if (!joint(safe)) { finish(); return; }
        bool blank = tok == c.blank_id;
        if (!blank && dur == 0 && t == last_emit_t && n_at_t >= 1) dur = 1;  // :318-323
        if (blank && dur == 0) dur = 1;                                       // :327-329
        int t_label = t;
        t += dur;
        safe = min(t, last);
        active = t < Teff;
        bool advance = active && blank;
        while (advance) {  // :348-405: same predictor state, blanks only move the frame pointer
            t_label = t;
            if (!joint(safe)) { finish(); return; }
            blank = tok == c.blank_id;
            if (blank && dur == 0) dur = 1;
            t += dur;
            safe = min(t, last);
            active = t < Teff;
            advance = active && blank;
        }
        if (active && tok != c.blank_id) {  // :409-463
            if (++processed > c.max_tokens_per_chunk) break;
            emit(t_label + goff);
            ++u;  // decoder LSTM step on the emitted token (:433-444)
            if (t_label == last_emit_t) ++n_at_t; else { last_emit_t = t_label; n_at_t = 1; }
            if (n_at_t >= c.max_symbols_per_step) {  // force-advance (:453-462)
                t = min(t + 1, last);
                safe = min(t, last);
                n_at_t = 0;
                last_emit_t = -1;
            }
        }
        active = t < Teff;
    }
    if (uni(a.is_last ? a.is_last[b] : 0)) {  // last-chunk flush (:472-571)
        int steps = 0, blanks = 0, fp = t;
        while (steps < c.max_symbols_per_step && blanks < c.consecutive_blank_limit) {
            const int sel = steps % 3;   // the three boundary-frame variants (:487-499) by selects: an indexed local array lives in scratch
            const int frame3 = sel == 0 ? min(fp, enc_len - 1) : (sel == 1 ? min(Teff - 1, enc_len - 1) : min(max(0, Teff - 2), enc_len - 1));
            if (!joint(frame3)) { finish(); return; }
            if (tok == c.blank_id) ++blanks;
            else {
                blanks = 0;
                emit(min(fp, Teff - 1) + goff);
                ++u;
            }
            fp = min(fp + max(1, dur), Teff);
            ++steps;
        }
    }
    if (writer) a.final_time[b] = t;
    finish();
}

// The same control loop for a WAVEFRONT that walks one chunk (tdt_logits_kernel), written as a state machine with ONE joint evaluation per
// iteration.  tdt_walk above calls the joint from three places (first decision of an outer step, the blank-advance loop, the last-chunk flush): inlined
// three times, with its state captured by reference, the compiler kept token / duration in scratch memory and the time / step counters in vector
// registers (every branch of the walk an exec-mask save + restore, every row address 64-bit per-lane arithmetic: the round-4 listing).  Here the state
// is a handful of integers forced uniform (readfirstlane) where they are loaded or decided: scalar registers, scalar branches, a scalar row address.
// `decide(u, frame, tok, bin)` -> one joint decision, `prob_of_last()` -> the probability of its token (asked for only when the token is emitted).
template <class Decide, class Prob>
__device__ __forceinline__ void tdt_walk_wave(const TdtArgs &a, const int b, const bool writer, Decide &&decide, Prob &&prob_of_last) {
    auto uni = [](const int x) { return __builtin_amdgcn_readfirstlane(x); };
    const fa_tdt_config &c = a.cfg;
    int32_t *otok = a.out_tok + static_cast<int64_t>(b) * a.max_out, *otime = a.out_time + static_cast<int64_t>(b) * a.max_out;
    int32_t *odur = a.out_dur + static_cast<int64_t>(b) * a.max_out;
    float *oconf = a.out_conf + static_cast<int64_t>(b) * a.max_out;
    int count = 0, st = FA_SUCCESS, u = 0;
    const int enc_len = uni(a.enc_len[b]);
    const int goff = uni(a.global_offset ? a.global_offset[b] : 0);
    const int emit_after = uni(a.emit_after ? a.emit_after[b] : -1);
    const int is_last = uni(a.is_last ? a.is_last[b] : 0);
    int t = uni(a.t0 ? a.t0[b] : 0);
    if (writer) a.final_time[b] = INT32_MIN;  // "timeJump not updated" (early returns, :110-112,:150-152)
    auto finish = [&]() { if (writer) { a.out_count[b] = count; a.final_u[b] = u; a.status[b] = st; } };
    if (enc_len <= 1) { finish(); return; }                        // :110-112
    const int Teff = min(enc_len, uni(a.audio_frames ? a.audio_frames[b] : enc_len));  // TdtFrameNavigation.swift:59-78
    if (t >= Teff) { finish(); return; }                           // :150-152
    const int last = Teff - 1;
    enum { OUTER = 0, INNER = 1, FLUSH = 2 };
    int phase = OUTER, last_emit_t = -1, n_at_t = 0, processed = 0, t_label = t;
    int steps = 0, blanks = 0, fp = 0;                             // last-chunk flush (:472-571)
    auto emit = [&](const int ts, const int tok, const int dur) {
        if (emit_after >= 0 && ts < emit_after) return;           // shouldEmitToken (:600-606)
        if (count < a.max_out) {
            const float score = clamp_probability(prob_of_last());
            if (writer) { otok[count] = tok; otime[count] = ts; odur[count] = dur; oconf[count] = score; }
        } else st = FA_OUTPUT_TOO_SMALL;
        ++count;
    };
    for (;;) {
        int frame;
        if (phase == FLUSH) {
            if (!(steps < c.max_symbols_per_step && blanks < c.consecutive_blank_limit)) break;
            const int sel = steps % 3;                             // the three boundary-frame variants (:487-499)
            frame = sel == 0 ? min(fp, enc_len - 1) : (sel == 1 ? min(Teff - 1, enc_len - 1) : min(max(0, Teff - 2), enc_len - 1));
        } else {
            if (phase == INNER) t_label = t;                       // :349: the blank-advance loop labels the frame it is about to look at
            frame = min(t, last);
        }
        // ---- one joint decision (the single call site)
        if (u >= a.U || frame < 0 || frame >= a.T) { st = FA_OUTPUT_TOO_SMALL; finish(); return; }
        int tok = 0, bi = 0;
        decide(u, frame, tok, bi);
        tok = uni(tok); bi = uni(bi);
        if (bi < 0 || bi >= c.n_duration_bins) { st = FA_RUNTIME_ERROR; finish(); return; }  // mapDurationBin throws (TdtDurationMapping.swift:17-22)
        int dur = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) dur = bi == q ? c.duration_bins[q] : dur;   // (selects on the uniform bin: an indexed copy of the config would sit in scratch)
        const bool blank = tok == c.blank_id;
        if (phase == FLUSH) {
            if (blank) ++blanks;
            else { blanks = 0; emit(min(fp, Teff - 1) + goff, tok, dur); ++u; }
            fp = min(fp + max(1, dur), Teff);
            ++steps;
            continue;
        }
        if (phase == OUTER) {
            if (!blank && dur == 0 && t == last_emit_t && n_at_t >= 1) dur = 1;  // :318-323
            t_label = t;
        }
        if (blank && dur == 0) dur = 1;                                           // :327-329 / :377-379
        t += dur;
        bool active = t < Teff;
        if (active && blank) { phase = INNER; continue; }                         // :348-405: same predictor state, blanks only move the frame pointer
        bool stop = false;
        if (active && !blank) {                                                   // :409-463
            if (++processed > c.max_tokens_per_chunk) stop = true;
            else {
                emit(t_label + goff, tok, dur);
                ++u;                                                              // decoder LSTM step on the emitted token (:433-444)
                if (t_label == last_emit_t) ++n_at_t; else { last_emit_t = t_label; n_at_t = 1; }
                if (n_at_t >= c.max_symbols_per_step) {                           // force-advance (:453-462)
                    t = min(t + 1, last);
                    n_at_t = 0;
                    last_emit_t = -1;
                }
            }
        }
        active = t < Teff;
        if (active && !stop) { phase = OUTER; continue; }
        if (!is_last) break;
        phase = FLUSH; steps = 0; blanks = 0; fp = t;
    }
    if (writer) a.final_time[b] = t;
    finish();
}

__global__ void tdt_kernel(const TdtArgs a) {   // joint decisions from tables [B][U][T], one thread per chunk
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int64_t tb = static_cast<int64_t>(b) * a.U * a.T;
    int64_t last = 0;
    tdt_walk(a, b, true, [&](const int u, const int frame, int &tok, int &bin) {
        last = tb + static_cast<int64_t>(u) * a.T + frame;
        tok = a.tok[last]; bin = a.bin[last];
    }, [&]() -> float { return a.prob[last]; });
}

// Joint decisions computed on the fly from joint LOGITS [B][U][T][row_stride] (token logits [0, V1), duration logits [V1, V1 + nd)):
// what the reference's JointDecision model hands back (TdtModelInference.swift:84-188: token_id, token_prob, duration) —
// token = first-index argmax over the V1 token logits (strict '>', NaN never wins: the rule of LogitsArgmax.swift:16-55),
// probability = softmax probability of that token, duration bin = first-index argmax over the nd duration logits.  One
// wavefront per chunk walks the greedy path and touches ONLY the rows on it (~T + tokens rows of V1 + nd logits) instead of the
// U x T x (V1 + nd) grid a table-building pre-pass would need.  The models themselves are not in the reference tree: PARITY UNPINNED.
struct TdtLogitArgs {
    const void *logits;
    int32_t f16, V1, nd;
    int64_t row_stride;
};

// Round 4: ONE WAVEFRONT per chunk (round 3: a 256-thread workgroup per chunk, three workgroup barriers and two passes over the row per
// decision — 4.2 us per decision, 3 % of the HBM roofline on 256 chunks).  A decision is latency, not work: ~4 KB of logits, an argmax and a
// soft-max denominator.  Within a wavefront both reductions stay in the VALU (DPP row shifts + row broadcasts; the first version used
// __shfl_xor: 36 LDS round trips per decision), the row is read ONCE (the denominator is accumulated against the running maximum, one
// rescale per batch of sixteen values), its loads are requested sixteen 256-byte pieces at a time and one batch ahead, and four times as
// many chunks fit a CU — the batch of chunks is the parallel axis of this kernel.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float tdt_dpp(const float old, const float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int tdt_dpp(const int old, const int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xf, false); }
// first maximum of (value, index) over the 64 lanes: lower index on equal values, NaN never present (callers keep it out); every lane gets the result
__device__ __forceinline__ void tdt_wave_argmax(float &v, int &i) {
#define FA_TDT_STEP(CTRL, MASK)                                                                   \
    { const float ov = tdt_dpp<CTRL, MASK>(-INFINITY, v); const int oi = tdt_dpp<CTRL, MASK>(0x7fffffff, i); \
      const bool t = (ov > v) | ((ov == v) & (oi < i)); v = t ? ov : v; i = t ? oi : i; }
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf) FA_TDT_STEP(0x118, 0xf)   // row_shr 1, 2, 4, 8: lane 15 of every row holds the row's result
    FA_TDT_STEP(0x142, 0xa) FA_TDT_STEP(0x143, 0xc)                                                   // row_bcast15 into rows 1, 3; row_bcast31 into rows 2, 3: lane 63 holds it all
#undef FA_TDT_STEP
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    i = __builtin_amdgcn_readlane(i, 63);
}
// the same over lanes 0 .. 7 only (values elsewhere are ignored): row_shr 1, 2, 4 leave the result in lane 7
__device__ __forceinline__ void tdt_low8_argmax(float &v, int &i) {
#define FA_TDT_STEP(CTRL, MASK)                                                                   \
    { const float ov = tdt_dpp<CTRL, MASK>(-INFINITY, v); const int oi = tdt_dpp<CTRL, MASK>(0x7fffffff, i); \
      const bool t = (ov > v) | ((ov == v) & (oi < i)); v = t ? ov : v; i = t ? oi : i; }
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf)
#undef FA_TDT_STEP
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 7));
    i = __builtin_amdgcn_readlane(i, 7);
}
// soft-max partials (m = maximum, s = sum of exp(x - m)) of the 64 lanes -> the wavefront's, in every lane: the maximum first (six DPP steps of
// one v_max), ONE rescale per lane, then the sum (six DPP adds) — the pairwise form spent two exp per step on the decision's dependent chain
__device__ __forceinline__ void tdt_wave_softmax(float &m, float &s) {
    float mm = m;
#define FA_TDT_STEP(CTRL, MASK) { const float om = tdt_dpp<CTRL, MASK>(-INFINITY, mm); mm = om > mm ? om : mm; }
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf) FA_TDT_STEP(0x118, 0xf)
    FA_TDT_STEP(0x142, 0xa) FA_TDT_STEP(0x143, 0xc)
#undef FA_TDT_STEP
    const float M = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mm), 63));
    float t = m == -INFINITY ? 0.0f : s * __expf(m - M);     // a lane without a finite value holds s = 0
#define FA_TDT_STEP(CTRL, MASK) t += tdt_dpp<CTRL, MASK>(0.0f, t);
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf) FA_TDT_STEP(0x118, 0xf)
    FA_TDT_STEP(0x142, 0xa) FA_TDT_STEP(0x143, 0xc)
#undef FA_TDT_STEP
    m = M;
    s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 63));
}

// ---- rows of at most 17 x 64 logits (Parakeet-TDT's 1 025 + 5, the CTC-sized heads): the row stays in registers, in its own element type, from
// the decision until the next one.  A decision is: 17 + 1 requests from a scalar row address (the lane's byte offsets are loop invariants; offsets
// beyond the row are clamped to its last element — a duplicate of a real element can tie with it but never beat it, and the true holder has the
// lower index), the lane maximum (v_max: a NaN operand is dropped), six DPP steps for the row maximum M, and the FIRST index holding M as 17
// compares whose lane masks the scalar unit searches (piece j before piece j + 1, lowest lane within a piece: index = lane + 64 j) — ~90 vector
// instructions where the (value, index) scan + (value, index) DPP reduction of round 4 took ~250.  The soft-max runs from the registers, and only
// when the token is emitted.
__device__ __forceinline__ float tdt_wave_max(float v) {
#define FA_TDT_STEP(CTRL, MASK) v = __builtin_fmaxf(v, tdt_dpp<CTRL, MASK>(-INFINITY, v));
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf) FA_TDT_STEP(0x118, 0xf)
    FA_TDT_STEP(0x142, 0xa) FA_TDT_STEP(0x143, 0xc)
#undef FA_TDT_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned tdt_wave_umin(unsigned v) {   // minimum over the 64 lanes, in every lane (uniform)
#define FA_TDT_STEP(CTRL, MASK) { const unsigned o = static_cast<unsigned>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), CTRL, MASK, 0xf, false)); v = o < v ? o : v; }
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf) FA_TDT_STEP(0x118, 0xf)
    FA_TDT_STEP(0x142, 0xa) FA_TDT_STEP(0x143, 0xc)
#undef FA_TDT_STEP
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ float tdt_low8_max(float v) {   // over lanes 0 .. 7
#define FA_TDT_STEP(CTRL, MASK) v = __builtin_fmaxf(v, tdt_dpp<CTRL, MASK>(-INFINITY, v));
    FA_TDT_STEP(0x111, 0xf) FA_TDT_STEP(0x112, 0xf) FA_TDT_STEP(0x114, 0xf)
#undef FA_TDT_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 7));
}

// W = logits per request and lane (round 6).  A request of one HALF per lane moves 128 bytes per wavefront, and with those the fp16 walk was SLOWER than
// the fp32 walk of the same chunks (0.179 against 0.155 ms for 1 024 chunks) although it reads half the bytes.  fp16 rows whose every start is 4-byte
// aligned are therefore read as PAIRS (W = 2: nine requests of 256 bytes instead of seventeen of 128; 0.142 ms, 4 096 chunks 33 -> 40 % of HBM): logit e
// of piece j of lane l is W (l + 64 j) + e.  The first logit of a piece is a token logit or a clamped duplicate of one (as above); the others are masked
// to -inf by their index when they lie behind the last token logit (a duration logit or the next row's first; the masks are loop invariants).  Pieces of
// 512 bytes and more per wavefront (fp32 pairs / quads, fp16 quads / octets) measured SLOWER than 256: profiles/r06_tdt_piece_probe.txt.
typedef unsigned tdt_v2u __attribute__((ext_vector_type(2)));
typedef unsigned tdt_v4u __attribute__((ext_vector_type(4)));
template <bool F16, int W = 1>
__global__ __launch_bounds__(64) void tdt_logits_fits_kernel(const TdtArgs a, const TdtLogitArgs g) {
    using E = std::conditional_t<F16, __half, float>;
    constexpr int kBytes = W * static_cast<int>(sizeof(E));   // per request and lane
    static_assert(W == 1 || kBytes == 4 || kBytes == 8 || kBytes == 16, "a request is one element or 1 / 2 / 4 dwords");
    constexpr int kReq = (17 + W - 1) / W;       // requests per lane: 64 W kReq >= 1 088 logits
    constexpr int kP = kReq * W;                 // logits per lane
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t tb = static_cast<int64_t>(b) * a.U * a.T;
    const int last_k = g.V1 - 1;
    auto index_of = [lane](const int j) { return W * (lane + 64 * (j / W)) + (j % W); };
    unsigned off[kReq];
#pragma unroll
    for (int j = 0; j < kReq; ++j) {
        const int k = W * (lane + 64 * j);
        const int lim = W == 1 ? last_k : last_k / W * W;                      // beyond the row: its last request again
        off[j] = static_cast<unsigned>(k < lim ? k : lim) * static_cast<unsigned>(sizeof(E));
    }
    const unsigned doff = static_cast<unsigned>(g.V1 + (lane < g.nd ? lane : g.nd - 1)) * static_cast<unsigned>(sizeof(E));
    const int row_bytes = (g.V1 + g.nd) * static_cast<int>(sizeof(E));
    float v[kP];   // fp16 rows are widened as they arrive (LogitsArgmax.swift:31-55 widens fp16 logits before the scan): one conversion per element —
                   // half-precision maxima (__hmax) compile to a NaN-handling branch per element
    tdt_walk_wave(a, b, lane == 0, [&](const int u, const int frame, int &tok, int &bin) {
        // the row through a buffer descriptor: scalar base (the walk's state is scalar) + the lane's 32-bit byte offset = ONE instruction per
        // request (a flat global load of scalar base + lane offset is compiled as a 64-bit per-lane add and the load)
        char *rp = const_cast<char *>(static_cast<const char *>(g.logits)) + (tb + static_cast<int64_t>(u) * a.T + frame) * g.row_stride * static_cast<int64_t>(sizeof(E));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rp, 0, row_bytes, 0x00020000);
        auto fetch = [&](const unsigned byte_off) -> float {
            if constexpr (F16) return __half2float(__ushort_as_half(__builtin_amdgcn_raw_buffer_load_b16(rsrc, static_cast<int>(byte_off), 0, 0)));
            else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, static_cast<int>(byte_off), 0, 0));
        };
        const float dve = fetch(doff);                                         // the duration logits travel with the row
        if constexpr (W == 1) {
#pragma unroll
            for (int j = 0; j < kReq; ++j) v[j] = fetch(off[j]);
        } else {
            constexpr int kDw = kBytes / 4;
            unsigned raw[kReq][kDw];
#pragma unroll
            for (int j = 0; j < kReq; ++j) {
                if constexpr (kDw == 1) raw[j][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, static_cast<int>(off[j]), 0, 0);
                else if constexpr (kDw == 2) { const tdt_v2u q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, static_cast<int>(off[j]), 0, 0); raw[j][0] = q.x; raw[j][1] = q.y; }
                else { const tdt_v4u q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(off[j]), 0, 0); raw[j][0] = q.x; raw[j][1] = q.y; raw[j][2] = q.z; raw[j][3] = q.w; }
            }
#pragma unroll
            for (int j = 0; j < kP; ++j) {
                float x;
                if constexpr (F16) {
                    const unsigned dw = raw[j / W][(j % W) / 2];
                    x = __half2float(__ushort_as_half(static_cast<unsigned short>((j & 1) ? dw >> 16 : dw & 0xffffu)));
                } else x = __uint_as_float(raw[j / W][j % W]);
                v[j] = (j % W == 0 || index_of(j) <= last_k) ? x : -INFINITY;
            }
        }
        float dv = lane < g.nd ? dve : -INFINITY;
        dv = dv != dv ? -INFINITY : dv;                                          // NaN never wins the first-maximum scan
        float me = v[0];
#pragma unroll
        for (int j = 1; j < kP; ++j) me = __builtin_fmaxf(me, v[j]);            // a NaN operand is dropped
        const float M = tdt_wave_max(me);                                         // all NaN: NaN; nothing above -inf: -inf
        // the FIRST index holding M: per lane the lowest of its logits that equals M (compares + selects, walked downwards so the lowest index is written
        // last), then the minimum of those indices over the wavefront — six v_min_u32 with a DPP operand.  Round 5 searched the 17 lane
        // masks on the scalar unit (ballot, compare, find-first, select per piece: 11 scalar instructions each, 318 per decision in r05_tdt_pmc.json).
        unsigned key = 0xffffffffu;
#pragma unroll
        for (int j = kP - 1; j >= 0; --j) key = v[j] == M ? static_cast<unsigned>(index_of(j)) : key;
        key = tdt_wave_umin(key);
        tok = (M > -INFINITY && key != 0xffffffffu) ? static_cast<int>(key) : 0;   // all NaN / -inf: index 0 (LogitsArgmax semantics: nothing beats the -inf seed)
        const float DM = tdt_low8_max(dv);
        const unsigned long long dmask = __builtin_amdgcn_ballot_w64(lane < g.nd && dv == DM);
        bin = (DM > -INFINITY && dmask != 0) ? __ffsll(static_cast<long long>(dmask)) - 1 : 0;   // first maximum; nothing above -inf: bin 0 (the scan's initial value)
    }, [&]() -> float {
        // soft-max of the row in the registers: lane maximum (NaN dropped), one exp per in-row value, then the wavefront's partials
        float m = -INFINITY, ssum = 0.0f;
        bool nan_seen = false;                     // a NaN logit anywhere in the row: probability 0 after the clamp
#pragma unroll
        for (int j = 0; j < kP; ++j) { const float x = v[j]; m = x > m ? x : m; nan_seen = nan_seen | (x != x); }
        const bool finite_max = m > -INFINITY;
#pragma unroll
        for (int j = 0; j < kP; ++j) { const float e = __expf(v[j] - m); ssum += (finite_max & (index_of(j) <= last_k)) ? e : 0.0f; }
        tdt_wave_softmax(m, ssum);
        return __builtin_amdgcn_ballot_w64(nan_seen) ? NAN : 1.0f / ssum;
    });
}

// ---- longer rows (8 198 logits: Parakeet-TDT v3): one pass, the soft-max partials (m, ssum) accumulated against the running maximum while the
// row streams through sixteen 256-byte pieces at a time, the NEXT sixteen requested before the present ones are looked at
template <bool F16>
__global__ __launch_bounds__(64) void tdt_logits_kernel(const TdtArgs a, const TdtLogitArgs g) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t tb = static_cast<int64_t>(b) * a.U * a.T;
    auto at = [&](const int64_t row, const int k) -> float {   // row: wave-uniform (a scalar base address), k: the lane's element
        if (F16) return __half2float((static_cast<const __half *>(g.logits) + row * g.row_stride)[k]);
        return (static_cast<const float *>(g.logits) + row * g.row_stride)[k];
    };
    constexpr int kBatch = 16;
    float m_on = -INFINITY, s_on = 0.0f;              // one-pass partials of the present row
    bool nan_seen = false;                            // a NaN logit anywhere in the row: probability 0 after the clamp (what the two-pass sum of round 3 gave)
    tdt_walk_wave(a, b, lane == 0, [&](const int u, const int frame, int &tok, int &bin) {
        const int64_t row = tb + static_cast<int64_t>(u) * a.T + frame;
        float dvl = lane < g.nd ? at(row, g.V1 + lane) : -INFINITY;            // the duration logits travel with the first batch of the row
        if (dvl != dvl) dvl = -INFINITY;                                       // NaN never wins the first-maximum scan
        // per-lane: first maximum over k = lane, lane + 64, ... (ascending, strict '>': NaN never wins) and the soft-max partials (m, ssum)
        float bv = -INFINITY, m = -INFINITY, ssum = 0.0f, v[kBatch], vn[kBatch];
        int bi = 0x7fffffff;
        nan_seen = false;
        auto request = [&](float (&dst)[kBatch], const int k0) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) { const int k = k0 + 64 * j; dst[j] = k < g.V1 ? at(row, k) : -INFINITY; }
        };
        request(v, lane);
        for (int k0 = lane; k0 < g.V1; k0 += 64 * kBatch) {
            request(vn, k0 + 64 * kBatch);                                          // (beyond the row: no loads, -inf)
            const int kbase = k0 - lane;                                             // wave-uniform: pieces at or beyond the row's end are skipped by a scalar branch
            float bm = -INFINITY;                                                    // maximum of the batch first: one exp per value, one rescale per batch
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {                                       // selects, no per-lane branches: the decision is a dependent chain
                if (kbase + 64 * j >= g.V1) break;
                const bool t = v[j] > bv;                                            // (slots beyond the row hold -inf and never win)
                bv = t ? v[j] : bv; bi = t ? k0 + 64 * j : bi;
                bm = v[j] > bm ? v[j] : bm;
            }
            {
                const bool up = bm > m;
                const float rescaled = m == -INFINITY ? 0.0f : ssum * __expf(m - bm);
                ssum = up ? rescaled : ssum; m = up ? bm : m;
            }
            const bool finite_max = m > -INFINITY;
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                if (kbase + 64 * j >= g.V1) break;
                const bool in_row = k0 + 64 * j < g.V1;
                nan_seen = nan_seen | (in_row & (v[j] != v[j]));
                const float e = __expf(v[j] - m);
                ssum += (in_row & finite_max) ? e : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) v[j] = vn[j];
        }
        m_on = m; s_on = ssum;
        tdt_wave_argmax(bv, bi);
        tok = bi == 0x7fffffff ? 0 : bi;           // all NaN / -inf: index 0 (LogitsArgmax semantics)
        float dv = dvl;
        int di = lane < g.nd ? lane : 0x7fffffff;
        tdt_low8_argmax(dv, di);                                                // the <= 8 duration logits sit in lanes 0 .. 7: three DPP steps
        bin = (di == 0x7fffffff || !(dv > -INFINITY)) ? 0 : di;                 // first maximum; nothing above -inf: bin 0 (the scan's initial value)
    }, [&]() -> float {
        float m = m_on, ssum = s_on;
        tdt_wave_softmax(m, ssum);
        return __builtin_amdgcn_ballot_w64(nan_seen) ? NAN : 1.0f / ssum;
    });
}

}  // namespace

extern "C" {

void fa_tdt_default_config(fa_tdt_config *c) {  // TdtConfig.swift:13-26
    if (!c) return;
    c->blank_id = 8192; c->max_symbols_per_step = 10; c->max_tokens_per_chunk = 150; c->consecutive_blank_limit = 5;
    c->n_duration_bins = 5;
    for (int i = 0; i < 8; ++i) c->duration_bins[i] = i < 5 ? i : 0;
}

int32_t fa_tdt_initial_time_index(int32_t has_time_jump, int32_t time_jump, int32_t context_frame_adjustment) {
    // TdtFrameNavigation.calculateInitialTimeIndices (TdtFrameNavigation.swift:20-49)
    if (!has_time_jump) return context_frame_adjustment;
    if (time_jump == 0 && context_frame_adjustment == 0) return kStandardOverlapFrames;
    const int32_t v = time_jump + context_frame_adjustment;
    return v > 0 ? v : 0;
}

void fa_tdt_navigation_state(int32_t time_indices, int32_t encoder_sequence_length, int32_t actual_audio_frames,
                             int32_t *effective_length, int32_t *safe_time_indices, int32_t *last_timestep, int32_t *active) {
    // TdtFrameNavigation.initializeNavigationState (:59-78)
    const int32_t eff = encoder_sequence_length < actual_audio_frames ? encoder_sequence_length : actual_audio_frames;
    if (effective_length) *effective_length = eff;
    if (safe_time_indices) *safe_time_indices = time_indices < eff - 1 ? time_indices : eff - 1;
    if (last_timestep) *last_timestep = eff - 1;
    if (active) *active = time_indices < eff;
}

int32_t fa_tdt_final_time_jump(int32_t current_time_indices, int32_t effective_length, int32_t is_last_chunk, int32_t *has_value) {
    // TdtFrameNavigation.calculateFinalTimeJump (:91-105): nil for the last chunk
    if (has_value) *has_value = !is_last_chunk;
    return is_last_chunk ? 0 : current_time_indices - effective_length;
}

fa_status fa_tdt_map_duration_bin(const fa_tdt_config *cfg, int32_t bin_index, int32_t *duration) {
    if (!cfg || !duration) return FA_INVALID_ARGUMENT;
    if (bin_index < 0 || bin_index >= cfg->n_duration_bins) return FA_RUNTIME_ERROR;  // "Duration bin index out of range" (:19-21)
    *duration = cfg->duration_bins[bin_index];
    return FA_SUCCESS;
}

float fa_tdt_clamp_probability(float v) { return clamp_probability(v); }

fa_status fa_tdt_greedy_tables_dev(fa_ctx *ctx, const fa_tdt_config *cfg, const int32_t *d_tok, const int32_t *d_bin, const float *d_prob,
                                   int32_t batch, int32_t U, int32_t T, const int32_t *d_enc_len, const int32_t *d_audio_frames,
                                   const int32_t *d_t0, const int32_t *d_is_last, const int32_t *d_global_offset,
                                   const int32_t *d_emit_after, int32_t max_out, int32_t *d_out_tok, int32_t *d_out_time,
                                   int32_t *d_out_dur, float *d_out_conf, int32_t *d_out_count, int32_t *d_final_time,
                                   int32_t *d_final_u, int32_t *d_status) {
    if (!ctx || !cfg) return FA_INVALID_ARGUMENT;
    if (batch == 0) return FA_SUCCESS;
    if (batch < 0 || U < 1 || T < 1 || max_out < 0 || cfg->n_duration_bins < 1 || cfg->n_duration_bins > 8 || !d_tok || !d_bin || !d_prob ||
        !d_enc_len || !d_out_count || !d_final_time || !d_final_u || !d_status || (max_out > 0 && (!d_out_tok || !d_out_time || !d_out_dur || !d_out_conf)))
        return fa::set_error(ctx, FA_INVALID_ARGUMENT, "tdt: bad arguments");
    fa::DeviceGuard guard(ctx->device);
    TdtArgs a;
    a.tok = d_tok; a.bin = d_bin; a.prob = d_prob; a.enc_len = d_enc_len; a.audio_frames = d_audio_frames; a.t0 = d_t0;
    a.is_last = d_is_last; a.global_offset = d_global_offset; a.emit_after = d_emit_after;
    a.out_tok = d_out_tok; a.out_time = d_out_time; a.out_dur = d_out_dur; a.out_conf = d_out_conf; a.out_count = d_out_count;
    a.final_time = d_final_time; a.final_u = d_final_u; a.status = d_status;
    a.B = batch; a.U = U; a.T = T; a.max_out = max_out; a.cfg = *cfg;
    hipLaunchKernelGGL(tdt_kernel, dim3((batch + 63) / 64), dim3(64), 0, ctx->stream, a);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

fa_status fa_tdt_greedy_logits_dev(fa_ctx *ctx, const fa_tdt_config *cfg, const void *d_logits, int32_t dtype, int32_t batch, int32_t U, int32_t T,
                                   int32_t vocab_with_blank, int64_t row_stride, const int32_t *d_enc_len, const int32_t *d_audio_frames,
                                   const int32_t *d_t0, const int32_t *d_is_last, const int32_t *d_global_offset, const int32_t *d_emit_after,
                                   int32_t max_out, int32_t *d_out_tok, int32_t *d_out_time, int32_t *d_out_dur, float *d_out_conf,
                                   int32_t *d_out_count, int32_t *d_final_time, int32_t *d_final_u, int32_t *d_status) {
    if (!ctx || !cfg) return FA_INVALID_ARGUMENT;
    if (batch == 0) return FA_SUCCESS;
    if (batch < 0 || U < 1 || T < 1 || max_out < 0 || cfg->n_duration_bins < 1 || cfg->n_duration_bins > 8 || vocab_with_blank < 1 ||
        row_stride < static_cast<int64_t>(vocab_with_blank) + cfg->n_duration_bins || (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_F16) || !d_logits ||
        !d_enc_len || !d_out_count || !d_final_time || !d_final_u || !d_status || (max_out > 0 && (!d_out_tok || !d_out_time || !d_out_dur || !d_out_conf)))
        return fa::set_error(ctx, FA_INVALID_ARGUMENT, "tdt: bad arguments");
    fa::DeviceGuard guard(ctx->device);
    TdtArgs a{};
    a.enc_len = d_enc_len; a.audio_frames = d_audio_frames; a.t0 = d_t0;
    a.is_last = d_is_last; a.global_offset = d_global_offset; a.emit_after = d_emit_after;
    a.out_tok = d_out_tok; a.out_time = d_out_time; a.out_dur = d_out_dur; a.out_conf = d_out_conf; a.out_count = d_out_count;
    a.final_time = d_final_time; a.final_u = d_final_u; a.status = d_status;
    a.B = batch; a.U = U; a.T = T; a.max_out = max_out; a.cfg = *cfg;
    TdtLogitArgs g{d_logits, dtype == FA_DTYPE_F16 ? 1 : 0, vocab_with_blank, cfg->n_duration_bins, row_stride};
    const bool fits = vocab_with_blank <= 64 * 17;   // the row stays in registers between its argmax and the (rare) request for its probability
    // fp16 rows that all start on a 4-byte boundary are read as pairs (tdt_logits_fits_kernel)
    const bool pairs = g.f16 && fits && (row_stride * 2) % 4 == 0 && reinterpret_cast<uintptr_t>(d_logits) % 4 == 0;
    const dim3 grid(batch), block(64);
    if (g.f16) {
        if (!fits) hipLaunchKernelGGL(tdt_logits_kernel<true>, grid, block, 0, ctx->stream, a, g);
        else if (pairs) hipLaunchKernelGGL((tdt_logits_fits_kernel<true, 2>), grid, block, 0, ctx->stream, a, g);
        else hipLaunchKernelGGL((tdt_logits_fits_kernel<true, 1>), grid, block, 0, ctx->stream, a, g);
    } else {
        if (!fits) hipLaunchKernelGGL(tdt_logits_kernel<false>, grid, block, 0, ctx->stream, a, g);
        else hipLaunchKernelGGL((tdt_logits_fits_kernel<false, 1>), grid, block, 0, ctx->stream, a, g);
    }
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

}  // extern "C"
