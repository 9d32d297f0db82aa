// ctc.hip — per-frame argmax + CTC greedy collapse for batches of logit matrices (gfx950).
//
// Replaces LogitsArgmax.argmaxPerFrame (reference: Sources/FluidAudio/ASR/Shared/LogitsArgmax.swift:16-55)
// and the collapse loop of ctcGreedyDecode (Sources/FluidAudio/ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:45-70,
// Sources/FluidAudio/ASR/SenseVoice/SenseVoiceManager.swift:119-126).
//
// One workgroup per matrix.  Each wavefront owns whole rows: 64 lanes stream the row with
// 16-byte loads, keep a (value, index) pair with the reference's strict '>' rule, and combine
// across lanes with xor-shuffles (greater value wins, equal values -> lower index, NaN never
// wins).  Row winners land in LDS; the collapse `id != blank && id != previous id` is a
// local predicate, so it becomes a block-wide exclusive scan + scatter.  Every logit is read
// from HBM exactly once: 4*T*V bytes per fp32 matrix.
#include <climits>

#include <hip/hip_fp16.h>

#include "fa_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kChunk = 2048;  // frames staged in LDS per pass (8 per thread in the scan)
constexpr int kPerThread = kChunk / kThreads;

struct CtcArgs {
    const void *logits;
    const int32_t *valid_frames;
    int32_t *frame_ids;
    int32_t *token_ids;
    int32_t *token_lens;
    int64_t row_stride, matrix_stride;
    int32_t frames, vocab, blank_id;
    int32_t vector_ok;  // rows are 16-byte aligned and vocab is a multiple of the vector width
    int32_t elem_ok;    // the matrices are aligned to their element type and a row holds two vectors or more: rows of any alignment take the head + 16-byte body + tail path
};


// Every logit is read once: a nontemporal 16-byte load keeps the stream out of the way of the L2 / MALL lines other kernels of the context own.
#ifndef FA_CTC_NT
#define FA_CTC_NT 1
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 stream_load(const uint4 *p) {
#if FA_CTC_NT
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ float4 stream_load(const float4 *p) {
#if FA_CTC_NT
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}

__device__ __forceinline__ void take(float x, int idx, float &best, int &bi) {
    const bool t = x > best;  // strict '>' => lowest index wins ties (CtcDecoder.swift:58-61); false for a NaN x, and best never becomes NaN
    best = t ? x : best;
    bi = t ? idx : bi;
}

__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        const bool t = (ov > v) | ((ov == v) & (oi < i));
        v = t ? ov : v;
        i = t ? oi : i;
    }
}

template <bool F16>
__device__ __forceinline__ int row_argmax(const void *row_ptr, const int vocab, const bool vector_ok, const int lane) {
    float best = -INFINITY;
    int bi = INT_MAX;
    if (F16) {
        const __half *row = static_cast<const __half *>(row_ptr);
        if (vector_ok) {
            const int nvec = vocab >> 3;
            const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
            if (lane < nvec) bi = lane * 8;
            for (int i = lane; i < nvec; i += 64) {
                const uint4 q = stream_load(r4 + i);
                const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    take(__half2float(__ushort_as_half(static_cast<unsigned short>(w[e] & 0xffffu))), 8 * i + 2 * e, best, bi);
                    take(__half2float(__ushort_as_half(static_cast<unsigned short>(w[e] >> 16))), 8 * i + 2 * e + 1, best, bi);
                }
            }
        } else {
            if (lane < vocab) bi = lane;
            for (int i = lane; i < vocab; i += 64) take(__half2float(row[i]), i, best, bi);
        }
    } else {
        const float *row = static_cast<const float *>(row_ptr);
        if (vector_ok) {
            const int nvec = vocab >> 2;
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            if (lane < nvec) bi = lane * 4;
            for (int i = lane; i < nvec; i += 64) {
                const float4 q = stream_load(r4 + i);
                take(q.x, 4 * i, best, bi);
                take(q.y, 4 * i + 1, best, bi);
                take(q.z, 4 * i + 2, best, bi);
                take(q.w, 4 * i + 3, best, bi);
            }
        } else {
            if (lane < vocab) bi = lane;
            for (int i = lane; i < vocab; i += 64) take(row[i], i, best, bi);
        }
    }
    wave_argmax(best, bi);
    return bi;  // all -inf/NaN row: every lane kept its first index, the minimum is 0
}

// R rows of one wavefront at a time (rows r, r + kWaves, ...): the R 16-byte loads of a step are issued back to back, so a wavefront keeps
// R times the bytes in flight; every row is still scanned in the order of row_argmax (same winner, same tie rule).
template <bool F16, int R>
__device__ __forceinline__ void rows_argmax(const char *row0, const size_t step_bytes, const int vocab, const int lane, int (&bi)[R]) {
    float best[R];
    const int nvec = F16 ? vocab >> 3 : vocab >> 2;
#pragma unroll
    for (int k = 0; k < R; ++k) { best[k] = -INFINITY; bi[k] = lane < nvec ? lane * (F16 ? 8 : 4) : INT_MAX; }
    for (int i = lane; i < nvec; i += 64) {
        uint4 q[R];
#pragma unroll
        for (int k = 0; k < R; ++k) q[k] = stream_load(reinterpret_cast<const uint4 *>(row0 + k * step_bytes) + i);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (F16) {
                    take(__half2float(__ushort_as_half(static_cast<unsigned short>(w[e] & 0xffffu))), 8 * i + 2 * e, best[k], bi[k]);
                    take(__half2float(__ushort_as_half(static_cast<unsigned short>(w[e] >> 16))), 8 * i + 2 * e + 1, best[k], bi[k]);
                } else {
                    take(__uint_as_float(w[e]), 4 * i + e, best[k], bi[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) wave_argmax(best[k], bi[k]);
}

// The same for rows that do NOT start on a 16-byte boundary or whose length is no multiple of the vector width — the common case in practice:
// Parakeet CTC logits are [T, 1025] (1 024 tokens + blank), so the alignment of a row rotates with its index.  A row is its head (the < 16 bytes up
// to the next boundary, lanes 0 ..), its 16-byte body with streaming loads, and its tail (< 16 bytes); a lane meets its elements in ascending index
// order, so the strict '>' keeps the first maximum as before.  (Until round 4 such rows took 4-byte loads, one row per wavefront.)
template <bool F16, int R>
__device__ __forceinline__ void rows_argmax_any(const char *row0, const size_t step_bytes, const int vocab, const int lane, int (&bi)[R]) {
    constexpr int VW = F16 ? 8 : 4, ESZ = F16 ? 2 : 4;
    // one element, unconverted: requested with a clamped index by EVERY lane (a load under a lane mask would be waited for on the spot)
    auto raw = [](const char *p, const int idx) -> unsigned {
        return F16 ? static_cast<unsigned>(reinterpret_cast<const unsigned short *>(p)[idx]) : reinterpret_cast<const unsigned *>(p)[idx];
    };
    auto value = [](const unsigned bits) -> float { return F16 ? __half2float(__ushort_as_half(static_cast<unsigned short>(bits))) : __uint_as_float(bits); };
    float best[R];
    unsigned hv[R], tv[R];
    int head[R], nvec[R], nv_max = 0;
    // the head and tail elements of all R rows are requested before anything is looked at
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const char *p = row0 + k * step_bytes;
        int h = static_cast<int>(((16u - static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) / ESZ);
        h = h < vocab ? h : vocab;
        head[k] = h; nvec[k] = (vocab - h) / VW;
        nv_max = nvec[k] > nv_max ? nvec[k] : nv_max;
        best[k] = -INFINITY; bi[k] = INT_MAX;
        hv[k] = raw(p, lane < h ? lane : 0);
        const int t0 = h + VW * nvec[k];
        tv[k] = raw(p, t0 + lane < vocab ? t0 + lane : vocab - 1);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) take(lane < head[k] ? value(hv[k]) : -INFINITY, lane, best[k], bi[k]);   // (-inf never wins: lanes without a head element keep their state)
    for (int i = lane; i < nv_max; i += 64) {
        uint4 q[R];
#pragma unroll
        for (int k = 0; k < R; ++k)   // clamped, not masked (nvec >= 1: vocab >= 2 VW on this path)
            q[k] = stream_load(reinterpret_cast<const uint4 *>(row0 + k * step_bytes + head[k] * ESZ) + (i < nvec[k] ? i : nvec[k] - 1));
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (i >= nvec[k]) continue;
            const unsigned w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
            const int base = head[k] + VW * i;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (F16) {
                    take(__half2float(__ushort_as_half(static_cast<unsigned short>(w[e] & 0xffffu))), base + 2 * e, best[k], bi[k]);
                    take(__half2float(__ushort_as_half(static_cast<unsigned short>(w[e] >> 16))), base + 2 * e + 1, best[k], bi[k]);
                } else {
                    take(__uint_as_float(w[e]), base + e, best[k], bi[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int ti = head[k] + VW * nvec[k] + lane;
        take(ti < vocab ? value(tv[k]) : -INFINITY, ti, best[k], bi[k]);
        wave_argmax(best[k], bi[k]);
        bi[k] = bi[k] == INT_MAX ? 0 : bi[k];   // nothing above -inf (or all NaN): index 0, as row_argmax
    }
}

#ifndef FA_CTC_ROWS
#define FA_CTC_ROWS 4  // A/B of 1 / 2 / 4 rows and of the load policy on MI355X: profiles/r04_ctc_ab.txt
#endif
constexpr int kRowsAtOnce = FA_CTC_ROWS;

// MODE 0: aligned rows of whole vectors; 1: rows of any alignment (head + body + tail); 2: 4-/2-byte loads (misaligned matrices, rows shorter than
// two vectors).  Separate builds: the any-alignment path needs 117 registers, the aligned one 74 — sharing a kernel cost it two wavefronts per SIMD
// (fp16: 82.6 -> 75.8 % of the HBM roofline).
template <bool F16, int MODE>
__global__ __launch_bounds__(kThreads) void ctc_greedy_kernel(const CtcArgs a) {
    __shared__ int32_t ids[kChunk];
    __shared__ int32_t wave_tot[kWaves];
    __shared__ int32_t carry_prev, carry_count;

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int T = a.frames;
    if (a.valid_frames) { const int v = a.valid_frames[b]; T = v < 0 ? 0 : (v < T ? v : T); }
    const size_t esz = F16 ? 2 : 4;
    const char *mat = static_cast<const char *>(a.logits) + static_cast<size_t>(b) * a.matrix_stride * esz;
    int32_t *out = a.token_ids + static_cast<int64_t>(b) * a.frames;
    int32_t *fids = a.frame_ids ? a.frame_ids + static_cast<int64_t>(b) * a.frames : nullptr;

    if (tid == 0) { carry_prev = -1; carry_count = 0; }
    __syncthreads();

    for (int c0 = 0; c0 < T; c0 += kChunk) {
        const int n = T - c0 < kChunk ? T - c0 : kChunk;
        // phase 1: one row per wavefront
        int r = wave;
        if (kRowsAtOnce > 1 && MODE == 0) {
            const size_t step_bytes = static_cast<size_t>(kWaves) * a.row_stride * esz;
            for (; r + (kRowsAtOnce - 1) * kWaves < n; r += kRowsAtOnce * kWaves) {
                int bi[kRowsAtOnce];
                rows_argmax<F16, kRowsAtOnce>(mat + static_cast<size_t>(c0 + r) * a.row_stride * esz, step_bytes, a.vocab, lane, bi);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < kRowsAtOnce; ++k) {
                        ids[r + k * kWaves] = bi[k];
                        if (fids) fids[c0 + r + k * kWaves] = bi[k];
                    }
                }
            }
        }
        if (kRowsAtOnce > 1 && MODE == 1) {
            const size_t step_bytes = static_cast<size_t>(kWaves) * a.row_stride * esz;
            for (; r + (kRowsAtOnce - 1) * kWaves < n; r += kRowsAtOnce * kWaves) {
                int bi[kRowsAtOnce];
                rows_argmax_any<F16, kRowsAtOnce>(mat + static_cast<size_t>(c0 + r) * a.row_stride * esz, step_bytes, a.vocab, lane, bi);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < kRowsAtOnce; ++k) {
                        ids[r + k * kWaves] = bi[k];
                        if (fids) fids[c0 + r + k * kWaves] = bi[k];
                    }
                }
            }
        }
        for (; r < n; r += kWaves) {
            const char *row = mat + static_cast<size_t>(c0 + r) * a.row_stride * esz;
            int bi;
            if (MODE == 1) { int one[1]; rows_argmax_any<F16, 1>(row, 0, a.vocab, lane, one); bi = one[0]; }
            else bi = row_argmax<F16>(row, a.vocab, MODE == 0, lane);
            if (lane == 0) {
                ids[r] = bi;
                if (fids) fids[c0 + r] = bi;
            }
        }
        __syncthreads();
        // phase 2: keep flags -> exclusive scan -> scatter (collapse, CtcDecoder.swift:52-68)
        const int prev0 = carry_prev, base = carry_count;
        int32_t mine[kPerThread];
        int cnt = 0;
        const int r0 = tid * kPerThread;
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int r = r0 + j;
            int keep = 0;
            int32_t id = 0;
            if (r < n) {
                id = ids[r];
                const int32_t prev = r > 0 ? ids[r - 1] : prev0;
                keep = (id != a.blank_id) && (id != prev);
            }
            mine[j] = keep ? id : -1;
            cnt += keep;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wave_base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int t = wave_tot[w];
            if (w < wave) wave_base += t;
            total += t;
        }
        int pos = base + wave_base + incl - cnt;
#pragma unroll
        for (int j = 0; j < kPerThread; ++j)
            if (mine[j] >= 0) out[pos++] = mine[j];
        __syncthreads();
        if (tid == 0) { carry_prev = ids[n - 1]; carry_count = base + total; }
        __syncthreads();
    }
    if (tid == 0) a.token_lens[b] = carry_count;
}

// ctcGreedyDecode(logProbs: [[Float]], ...) (CtcDecoder.swift:15-36) is NOT the [1, T, V] overload with another container: it seeds the scan with
// frame[0] (`bestVal = frame[0]`, :25), so a frame whose element 0 is NaN decodes to index 0 (no later `frame[v] > NaN` is ever true), where the
// -inf seed of :55-64 skips the NaN and finds the finite maximum; every frame has its own length (`1..<frame.count`, :26); an empty frame is skipped
// BEFORE `prev` is touched (`guard !frame.isEmpty else { continue }`, :23), so `a, [], a` collapses to one `a`.  For frames whose element 0 is not NaN
// the two seeds agree (a non-NaN frame[0] beats or equals -inf; later NaNs lose either way; all -inf -> 0 in both).
//
// One workgroup per utterance, one wavefront per frame; the frames of utterance u are rows utt_rows[u] .. utt_rows[u + 1] of a flat value array whose
// row r spans values[row_offsets[r] .. row_offsets[r + 1]).  Frames of eight values or more go through the head + 16-byte body + tail scan above.
constexpr int32_t kEmptyFrame = INT_MIN;   // LDS marker of an empty frame (a frame id is >= 0)

struct CtcRowsArgs {
    const float *values;
    const int64_t *row_offsets;   // [total_rows + 1], non-decreasing
    const int64_t *utt_rows;      // [batch + 1], non-decreasing; NULL: one utterance of `total_rows` rows
    int32_t *frame_ids;           // [total_rows] or NULL: argmax per frame, -1 for an empty frame
    int32_t *token_ids;           // [total_rows]: utterance u writes from token_ids[utt_rows[u]]
    int32_t *token_lens;          // [batch]
    int64_t total_rows;
    int32_t blank_id;
};

__device__ __forceinline__ int row_argmax_seed_first(const float *row, const int64_t len, const int lane) {
    int bi;
    if (len >= 8 && len <= INT_MAX) {
        int one[1];
        rows_argmax_any<false, 1>(reinterpret_cast<const char *>(row), 0, static_cast<int>(len), lane, one);
        bi = one[0];
    } else {
        float best = -INFINITY;
        int64_t b64 = INT64_MAX;
        for (int64_t i = lane; i < len; i += 64) { const float x = row[i]; const bool t = x > best; best = t ? x : best; b64 = t ? i : b64; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int64_t oi = __shfl_xor(b64, off);
            const bool t = (ov > best) | ((ov == best) & (oi < b64));
            best = t ? ov : best; b64 = t ? oi : b64;
        }
        bi = b64 == INT64_MAX ? 0 : static_cast<int>(b64 > INT_MAX ? INT_MAX : b64);
    }
    const float first = row[0];                 // uniform address: one scalar-like broadcast load
    return first != first ? 0 : bi;             // `bestVal = frame[0]` (:25): a NaN seed is never beaten
}

__global__ __launch_bounds__(kThreads) void ctc_greedy_rows_kernel(const CtcRowsArgs a) {
    __shared__ int32_t ids[kChunk];
    __shared__ int32_t wave_tot[kWaves], wave_last[kWaves];
    __shared__ int32_t carry_prev, carry_count;

    const int u = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row_lo = a.utt_rows ? a.utt_rows[u] : 0, row_hi = a.utt_rows ? a.utt_rows[u + 1] : a.total_rows;
    const int64_t T = row_hi - row_lo;
    int32_t *out = a.token_ids + row_lo;
    int32_t *fids = a.frame_ids ? a.frame_ids + row_lo : nullptr;

    if (tid == 0) { carry_prev = -1; carry_count = 0; }
    __syncthreads();

    for (int64_t c0 = 0; c0 < T; c0 += kChunk) {
        const int n = T - c0 < kChunk ? static_cast<int>(T - c0) : kChunk;
        for (int r = wave; r < n; r += kWaves) {
            const int64_t lo = a.row_offsets[row_lo + c0 + r], hi = a.row_offsets[row_lo + c0 + r + 1];
            int id = kEmptyFrame;
            if (hi > lo) id = row_argmax_seed_first(a.values + lo, hi - lo, lane);
            if (lane == 0) {
                ids[r] = id;
                if (fids) fids[c0 + r] = id == kEmptyFrame ? -1 : id;
            }
        }
        __syncthreads();
        // `prev` of a frame = the id of the nearest non-empty frame before it: a "rightmost non-empty" scan next to the count scan
        const int prev0 = carry_prev, base = carry_count;
        int32_t mine[kPerThread];
        int cnt = 0;
        const int r0 = tid * kPerThread;
        int32_t seg_last = kEmptyFrame;
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int r = r0 + j;
            mine[j] = r < n ? ids[r] : kEmptyFrame;
            seg_last = mine[j] != kEmptyFrame ? mine[j] : seg_last;
        }
        int32_t last_incl = seg_last;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t o = __shfl_up(last_incl, off);
            if (lane >= off && last_incl == kEmptyFrame) last_incl = o;
        }
        if (lane == 63) wave_last[wave] = last_incl;
        int32_t prev = __shfl_up(last_incl, 1);
        if (lane == 0) prev = kEmptyFrame;
        __syncthreads();
        if (prev == kEmptyFrame) {
            for (int w = wave - 1; w >= 0 && prev == kEmptyFrame; --w) prev = wave_last[w];
            if (prev == kEmptyFrame) prev = prev0;
        }
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int32_t id = mine[j];
            const int keep = (id != kEmptyFrame) && (id != a.blank_id) && (id != prev);
            prev = id != kEmptyFrame ? id : prev;
            mine[j] = keep ? id : -1;
            cnt += keep;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wave_base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int t = wave_tot[w];
            if (w < wave) wave_base += t;
            total += t;
        }
        int pos = base + wave_base + incl - cnt;
#pragma unroll
        for (int j = 0; j < kPerThread; ++j)
            if (mine[j] >= 0) out[pos++] = mine[j];
        __syncthreads();
        if (tid == 0) {
            int32_t last = kEmptyFrame;
            for (int w = kWaves - 1; w >= 0 && last == kEmptyFrame; --w) last = wave_last[w];
            carry_prev = last == kEmptyFrame ? prev0 : last;
            carry_count = base + total;
        }
        __syncthreads();
    }
    if (tid == 0) a.token_lens[u] = carry_count;
}

// Per-frame log-softmax with temperature and blank bias as CtcKeywordSpotter.makeLogProbs / logSoftmax do it
// (reference: Sources/FluidAudio/ASR/Parakeet/SlidingWindow/CustomVocabulary/WordSpotting/CtcKeywordSpotter+Inference.swift:350-431):
// x / temperature (only when temperature != 1), max, sum of expf(x - max), (x - max) - logf(sum); then blankBias is
// subtracted from the blank column.  One wavefront per row; the row is read ONCE from HBM and lives in registers
// (up to 64 lanes x 32 values), so the kernel moves 2 x 4 T V bytes per matrix.
constexpr int kLsmRegs = 32;  // values per lane held in registers: V <= 2048 single pass

struct LsmArgs {
    const void *logits;
    float *out;
    int64_t row_stride, matrix_stride, out_row_stride, out_matrix_stride, rows_total;
    int32_t frames, vocab, blank_id;
    float inv_temp_unused, temperature, blank_bias;
};

template <bool F16>
__device__ __forceinline__ float lsm_load(const void *row, const int i) {
    return F16 ? __half2float(static_cast<const __half *>(row)[i]) : static_cast<const float *>(row)[i];
}

template <bool F16>
__global__ __launch_bounds__(kThreads) void ctc_log_softmax_kernel(const LsmArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
    if (r >= a.rows_total) return;
    const int64_t b = r / a.frames, t = r % a.frames;
    const size_t esz = F16 ? 2 : 4;
    const char *row = static_cast<const char *>(a.logits) + (static_cast<size_t>(b) * a.matrix_stride + static_cast<size_t>(t) * a.row_stride) * esz;
    float *orow = a.out + b * a.out_matrix_stride + t * a.out_row_stride;
    const int V = a.vocab;
    const bool scale = a.temperature != 1.0f;
    float x[kLsmRegs];
    float mx = -INFINITY;
    const bool in_regs = V <= 64 * kLsmRegs;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kLsmRegs; ++j) {
            const int i = lane + 64 * j;
            float v = -INFINITY;
            if (i < V) { v = lsm_load<F16>(row, i); if (scale) v = v / a.temperature; }
            x[j] = v;
            mx = fmaxf(mx, v);
        }
    } else {
        for (int i = lane; i < V; i += 64) { float v = lsm_load<F16>(row, i); if (scale) v = v / a.temperature; mx = fmaxf(mx, v); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kLsmRegs; ++j) if (lane + 64 * j < V) sum += expf(x[j] - mx);
    } else {
        for (int i = lane; i < V; i += 64) { float v = lsm_load<F16>(row, i); if (scale) v = v / a.temperature; sum += expf(v - mx); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float lse = logf(sum);
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kLsmRegs; ++j) {
            const int i = lane + 64 * j;
            if (i < V) { float o = (x[j] - mx) - lse; if (i == a.blank_id && a.blank_bias != 0.0f) o -= a.blank_bias; orow[i] = o; }
        }
    } else {
        for (int i = lane; i < V; i += 64) {
            float v = lsm_load<F16>(row, i); if (scale) v = v / a.temperature;
            float o = (v - mx) - lse; if (i == a.blank_id && a.blank_bias != 0.0f) o -= a.blank_bias; orow[i] = o;
        }
    }
}

// fp32 rows that are 16-byte aligned with V a multiple of 4 and V <= 2048: the same computation with 16-byte loads and stores
// (lane l owns the float4 groups l, l + 64, ...: 4x fewer memory instructions than the strided scalar form)
__global__ __launch_bounds__(kThreads) void ctc_log_softmax_vec4_kernel(const LsmArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
    if (r >= a.rows_total) return;
    const int64_t b = r / a.frames, t = r % a.frames;
    const float4 *row = reinterpret_cast<const float4 *>(static_cast<const float *>(a.logits) + b * a.matrix_stride + t * a.row_stride);
    float4 *orow = reinterpret_cast<float4 *>(a.out + b * a.out_matrix_stride + t * a.out_row_stride);
    const int nvec = a.vocab >> 2;
    const bool scale = a.temperature != 1.0f;
    constexpr int kVecRegs = kLsmRegs / 4;
    float4 x[kVecRegs];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kVecRegs; ++j) {
        const int i = lane + 64 * j;
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (i < nvec) {
            v = row[i];
            if (scale) { v.x = v.x / a.temperature; v.y = v.y / a.temperature; v.z = v.z / a.temperature; v.w = v.w / a.temperature; }
        }
        x[j] = v;
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecRegs; ++j)
        if (lane + 64 * j < nvec) sum += (expf(x[j].x - mx) + expf(x[j].y - mx)) + (expf(x[j].z - mx) + expf(x[j].w - mx));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float lse = logf(sum);
    const bool bias = a.blank_bias != 0.0f && a.blank_id >= 0;
#pragma unroll
    for (int j = 0; j < kVecRegs; ++j) {
        const int i = lane + 64 * j;
        if (i >= nvec) continue;
        float4 o = make_float4((x[j].x - mx) - lse, (x[j].y - mx) - lse, (x[j].z - mx) - lse, (x[j].w - mx) - lse);
        if (bias && (a.blank_id >> 2) == i) {
            const int c = a.blank_id & 3;
            if (c == 0) o.x -= a.blank_bias; else if (c == 1) o.y -= a.blank_bias; else if (c == 2) o.z -= a.blank_bias; else o.w -= a.blank_bias;
        }
        orow[i] = o;
    }
}

fa_status check_args(fa_ctx *ctx, const void *logits, int dtype, int batch, int frames, int vocab, int64_t row_stride,
                     int64_t matrix_stride, const int32_t *token_ids, const int32_t *token_lens) {
    if (!ctx || !token_ids || !token_lens) return FA_INVALID_ARGUMENT;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_F16) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc: bad dtype");
    if (batch < 0 || frames < 0 || vocab < 1 || row_stride < vocab) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc: bad shape");
    if (batch > 0 && frames > 0 && (!logits || matrix_stride < static_cast<int64_t>(frames - 1) * row_stride + vocab))
        return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc: bad strides");
    return FA_SUCCESS;
}

}  // namespace

extern "C" {

fa_status fa_ctc_greedy_batch_dev(fa_ctx *ctx, const void *d_logits, int32_t dtype, int32_t batch, int32_t frames,
                                  int32_t vocab, int64_t row_stride, int64_t matrix_stride,
                                  const int32_t *d_valid_frames, int32_t blank_id, int32_t *d_frame_ids,
                                  int32_t *d_token_ids, int32_t *d_token_lens) {
    FA_TRY(check_args(ctx, d_logits, dtype, batch, frames, vocab, row_stride, matrix_stride, d_token_ids, d_token_lens));
    if (batch == 0) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    CtcArgs a;
    a.logits = d_logits; a.valid_frames = d_valid_frames; a.frame_ids = d_frame_ids; a.token_ids = d_token_ids;
    a.token_lens = d_token_lens; a.row_stride = row_stride; a.matrix_stride = matrix_stride; a.frames = frames;
    a.vocab = vocab; a.blank_id = blank_id;
    const int vw = dtype == FA_DTYPE_F16 ? 8 : 4;
    a.vector_ok = (vocab % vw == 0) && (row_stride % vw == 0) && (matrix_stride % vw == 0) &&
                  (reinterpret_cast<uintptr_t>(d_logits) % 16 == 0);
    a.elem_ok = reinterpret_cast<uintptr_t>(d_logits) % (dtype == FA_DTYPE_F16 ? 2 : 4) == 0 && vocab >= 2 * vw;   // at least one whole 16-byte piece behind any head
    const int launch_mode = a.vector_ok ? 0 : (a.elem_ok ? 1 : 2);
    const bool f16 = dtype == FA_DTYPE_F16;
#define FA_CTC_LAUNCH(F, M) hipLaunchKernelGGL((ctc_greedy_kernel<F, M>), dim3(batch), dim3(kThreads), 0, ctx->stream, a)
    if (launch_mode == 0) { if (f16) FA_CTC_LAUNCH(true, 0); else FA_CTC_LAUNCH(false, 0); }
    else if (launch_mode == 1) { if (f16) FA_CTC_LAUNCH(true, 1); else FA_CTC_LAUNCH(false, 1); }
    else { if (f16) FA_CTC_LAUNCH(true, 2); else FA_CTC_LAUNCH(false, 2); }
#undef FA_CTC_LAUNCH
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

fa_status fa_ctc_greedy_batch(fa_ctx *ctx, const void *logits, int32_t dtype, int32_t batch, int32_t frames, int32_t vocab,
                              int64_t row_stride, int64_t matrix_stride, const int32_t *valid_frames, int32_t blank_id,
                              int32_t *frame_ids, int32_t *token_ids, int32_t *token_lens) {
    FA_TRY(check_args(ctx, logits, dtype, batch, frames, vocab, row_stride, matrix_stride, token_ids, token_lens));
    if (batch == 0) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    const size_t esz = dtype == FA_DTYPE_F16 ? 2 : 4;
    const size_t in_bytes = frames > 0 ? (static_cast<size_t>(batch - 1) * matrix_stride + static_cast<size_t>(frames - 1) * row_stride + vocab) * esz : 0;
    const size_t id_bytes = sizeof(int32_t) * static_cast<size_t>(batch) * (frames > 0 ? frames : 1);
    fa::DevBuf d_in, d_valid, d_fid, d_tok, d_len;
    hipError_t e;
    fa_status st = FA_SUCCESS;
    do {
        if ((e = d_in.alloc(in_bytes)) != hipSuccess) break;
        if ((e = d_tok.alloc(id_bytes)) != hipSuccess) break;
        if ((e = d_len.alloc(sizeof(int32_t) * batch)) != hipSuccess) break;
        if (frame_ids && (e = d_fid.alloc(id_bytes)) != hipSuccess) break;
        if (valid_frames && (e = d_valid.alloc(sizeof(int32_t) * batch)) != hipSuccess) break;
        if (in_bytes && (e = hipMemcpyAsync(d_in.p, logits, in_bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if (valid_frames && (e = hipMemcpyAsync(d_valid.p, valid_frames, sizeof(int32_t) * batch, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        st = fa_ctc_greedy_batch_dev(ctx, d_in.p, dtype, batch, frames, vocab, row_stride, matrix_stride,
                                     valid_frames ? d_valid.as<int32_t>() : nullptr, blank_id,
                                     frame_ids ? d_fid.as<int32_t>() : nullptr, d_tok.as<int32_t>(), d_len.as<int32_t>());
        if (st != FA_SUCCESS) break;
        if (frames > 0 && (e = hipMemcpyAsync(token_ids, d_tok.p, id_bytes, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        if (frames > 0 && frame_ids && (e = hipMemcpyAsync(frame_ids, d_fid.p, id_bytes, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(token_lens, d_len.p, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    if (st != FA_SUCCESS) return st;
    return fa::hip_status(ctx, e, "fa_ctc_greedy_batch");
}

fa_status fa_ctc_greedy_rows_dev(fa_ctx *ctx, const float *d_values, const int64_t *d_row_offsets, int64_t total_rows, const int64_t *d_utt_rows,
                                 int32_t batch, int32_t blank_id, int32_t *d_frame_ids, int32_t *d_token_ids, int32_t *d_token_lens) {
    if (!ctx || !d_token_lens) return FA_INVALID_ARGUMENT;
    if (batch < 0 || total_rows < 0) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: bad shape");
    if (batch == 0) return FA_SUCCESS;
    if (!d_utt_rows && batch != 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: a batch needs utt_rows");
    if (total_rows > 0 && (!d_row_offsets || !d_token_ids)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: null buffer");
    fa::DeviceGuard guard(ctx->device);
    CtcRowsArgs a;
    a.values = d_values; a.row_offsets = d_row_offsets; a.utt_rows = d_utt_rows; a.frame_ids = d_frame_ids; a.token_ids = d_token_ids;
    a.token_lens = d_token_lens; a.total_rows = total_rows; a.blank_id = blank_id;
    hipLaunchKernelGGL(ctc_greedy_rows_kernel, dim3(batch), dim3(kThreads), 0, ctx->stream, a);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

fa_status fa_ctc_greedy_rows(fa_ctx *ctx, const float *values, const int64_t *row_offsets, int64_t total_rows, const int64_t *utt_rows, int32_t batch,
                             int32_t blank_id, int32_t *frame_ids, int32_t *token_ids, int32_t *token_lens) {
    if (!ctx || !token_lens) return FA_INVALID_ARGUMENT;
    if (batch < 0 || total_rows < 0) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: bad shape");
    if (batch == 0) return FA_SUCCESS;
    if (!utt_rows && batch != 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: a batch needs utt_rows");
    if (total_rows > 0 && (!row_offsets || !token_ids)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: null buffer");
    // the offsets are the caller's: a decreasing pair would make the kernel read outside `values`
    if (total_rows > 0) {
        if (row_offsets[0] < 0) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: negative offset");
        for (int64_t r = 0; r < total_rows; ++r)
            if (row_offsets[r + 1] < row_offsets[r]) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: row_offsets decrease");
    }
    if (utt_rows) {
        if (utt_rows[0] < 0 || utt_rows[batch] > total_rows) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: utt_rows out of range");
        for (int32_t u = 0; u < batch; ++u)
            if (utt_rows[u + 1] < utt_rows[u]) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: utt_rows decrease");
    }
    const int64_t n_values = total_rows > 0 ? row_offsets[total_rows] : 0;
    if (n_values > 0 && !values) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "ctc rows: null values");
    fa::DeviceGuard guard(ctx->device);
    const size_t id_bytes = sizeof(int32_t) * static_cast<size_t>(total_rows > 0 ? total_rows : 1);
    fa::DevBuf d_val, d_off, d_utt, d_fid, d_tok, d_len;
    hipError_t e;
    fa_status st = FA_SUCCESS;
    do {
        if ((e = d_val.alloc(sizeof(float) * static_cast<size_t>(n_values > 0 ? n_values : 1))) != hipSuccess) break;
        if ((e = d_off.alloc(sizeof(int64_t) * static_cast<size_t>(total_rows + 1))) != hipSuccess) break;
        if ((e = d_tok.alloc(id_bytes)) != hipSuccess) break;
        if ((e = d_len.alloc(sizeof(int32_t) * batch)) != hipSuccess) break;
        if (frame_ids && (e = d_fid.alloc(id_bytes)) != hipSuccess) break;
        if (utt_rows && (e = d_utt.alloc(sizeof(int64_t) * (static_cast<size_t>(batch) + 1))) != hipSuccess) break;
        if (n_values > 0 && (e = hipMemcpyAsync(d_val.p, values, sizeof(float) * static_cast<size_t>(n_values), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if (total_rows > 0 && (e = hipMemcpyAsync(d_off.p, row_offsets, sizeof(int64_t) * static_cast<size_t>(total_rows + 1), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if (utt_rows && (e = hipMemcpyAsync(d_utt.p, utt_rows, sizeof(int64_t) * (static_cast<size_t>(batch) + 1), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        st = fa_ctc_greedy_rows_dev(ctx, d_val.as<float>(), d_off.as<int64_t>(), total_rows, utt_rows ? d_utt.as<int64_t>() : nullptr, batch, blank_id,
                                    frame_ids ? d_fid.as<int32_t>() : nullptr, d_tok.as<int32_t>(), d_len.as<int32_t>());
        if (st != FA_SUCCESS) break;
        if (total_rows > 0 && (e = hipMemcpyAsync(token_ids, d_tok.p, sizeof(int32_t) * static_cast<size_t>(total_rows), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        if (total_rows > 0 && frame_ids && (e = hipMemcpyAsync(frame_ids, d_fid.p, sizeof(int32_t) * static_cast<size_t>(total_rows), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(token_lens, d_len.p, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    if (st != FA_SUCCESS) return st;
    return fa::hip_status(ctx, e, "fa_ctc_greedy_rows");
}

fa_status fa_ctc_log_softmax_batch_dev(fa_ctx *ctx, const void *d_logits, int32_t dtype, int32_t batch, int32_t frames, int32_t vocab,
                                       int64_t row_stride, int64_t matrix_stride, float temperature, float blank_bias, int32_t blank_id,
                                       float *d_log_probs) {
    if (!ctx || !d_log_probs) return FA_INVALID_ARGUMENT;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_F16) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "log_softmax: bad dtype");
    if (batch < 0 || frames < 0 || vocab < 1 || row_stride < vocab || !(temperature > 0.0f)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "log_softmax: bad shape");
    if (batch == 0 || frames == 0) return FA_SUCCESS;
    if (!d_logits || matrix_stride < static_cast<int64_t>(frames - 1) * row_stride + vocab) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "log_softmax: bad strides");
    fa::DeviceGuard guard(ctx->device);
    LsmArgs a{};
    a.logits = d_logits; a.out = d_log_probs; a.row_stride = row_stride; a.matrix_stride = matrix_stride;
    a.out_row_stride = vocab; a.out_matrix_stride = static_cast<int64_t>(frames) * vocab;
    a.rows_total = static_cast<int64_t>(batch) * frames; a.frames = frames; a.vocab = vocab; a.blank_id = blank_id;
    a.temperature = temperature; a.blank_bias = blank_bias;
    const unsigned grid = static_cast<unsigned>((a.rows_total + kWaves - 1) / kWaves);
    const bool vec4 = dtype == FA_DTYPE_F32 && vocab % 4 == 0 && vocab <= 64 * kLsmRegs && row_stride % 4 == 0 && matrix_stride % 4 == 0 &&
                      reinterpret_cast<uintptr_t>(d_logits) % 16 == 0 && reinterpret_cast<uintptr_t>(d_log_probs) % 16 == 0;
    if (vec4) hipLaunchKernelGGL(ctc_log_softmax_vec4_kernel, dim3(grid), dim3(kThreads), 0, ctx->stream, a);
    else if (dtype == FA_DTYPE_F16) hipLaunchKernelGGL(ctc_log_softmax_kernel<true>, dim3(grid), dim3(kThreads), 0, ctx->stream, a);
    else hipLaunchKernelGGL(ctc_log_softmax_kernel<false>, dim3(grid), dim3(kThreads), 0, ctx->stream, a);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

}  // extern "C"
