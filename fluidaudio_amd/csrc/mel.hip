// mel.hip — batched STFT -> power -> Slaney mel -> log featurizer for gfx950.
//
// Replaces AudioMelSpectrogram.computeFlat / computeFlatTransposed / compute
// (reference: Sources/FluidAudio/Shared/AudioMelSpectrogram.swift:132-178,185-292,325-456).
//
// One workgroup (256 threads = 4 wavefronts) walks a contiguous range of 32-frame tiles.  Per tile the
// hop*31+512 pre-emphasised samples are staged in LDS once; then every wavefront runs its own 8 frames
// (two passes of 4 frames x 16 lanes) with NO workgroup barrier: window and twiddles sit in registers,
// the only LDS traffic of the FFT is one radix-16 transpose, the Z[k] <-> Z[256-k] exchange of the
// real-FFT recombination is a DPP lane permutation, and the 257 power bins go back to the wavefront's
// LDS region only to be gathered by the sparse triangular filterbank.  Log-mel values are staged in LDS
// and stored by the whole workgroup as full rows ([n_mels, T]: 128-byte runs; [T, n_mels]: contiguous).
// HBM traffic per 15 s utterance = 960 000 B read + 768 512 B written.
#include <algorithm>
#include <climits>
#include <cmath>
#include <mutex>
#include <vector>

#include "fa_common.h"
#include "mel_core.h"
#include "mel_generic.h"
#include "mel_pk.h"

using namespace fa::melcore;
using fa::melpk::f2;

namespace {

constexpr int kTileFrames = 32;
constexpr int kThreads = 256;
constexpr int kWaveFrames = 4;                 // frames in flight per wavefront (16 lanes each)
constexpr int kRegions = kThreads / kGroup;    // 16 LDS regions (4 waves x 4 frames)
constexpr int kPasses = kTileFrames / kRegions;  // 2
constexpr int kMaxMels = 256;
constexpr int kMelPad = 36;                    // [n_mels][36] staging: rows 16-byte aligned for float4 reads
constexpr int kFramePad = 8;                   // [32][n_mels + 8] staging

struct MelArgs {
    const float *pcm;
    const int64_t *offsets;   // B+1
    const int32_t *frames;    // B   (T per utterance)
    const float *last;        // B or nullptr
    float *out;
    int32_t *lengths;         // B or nullptr
    const float *windowz;     // 512  window zero-extended to n_fft at offset `off`
    const float2 *tw256;      // 256  exp(-2 pi i k / 256)
    const float2 *tw512;      // 129  exp(-2 pi i k / 512)
    const int32_t *mel_tab;   // n_mels packed (lo | cnt<<10 | start<<20)
    const float *mel_w;       // n_weights
    int64_t utt_stride;
    int64_t total_tiles;
    int32_t tiles_per_utt;
    int32_t frame_stride;
    int32_t n_mels, n_weights;
    int32_t hop, pad, stage_count, stage_alloc, out_alloc;
    float preemph, log_floor;
    int32_t floor_clamped;
    unsigned long long *queue;       // mel_kernel_v4: tile counter (never reset) ...
    unsigned long long queue_base;   // ... and the first value that belongs to this launch
    unsigned long long *prof;  // FA_MEL_PROF env (diagnostics): per-phase cycle sums of one workgroup's wave 0
    int32_t prio_lo, prio_hi, prio_pw, prio_rd;  // wave priorities: FFT / filterbank..store / power / sample reads (FA_MEL_PRIO=a,b,c,d)
};

// lane l <- lane (16 - l) & 15 inside every row of 16 lanes: row_mirror (l <- 15 - l), then row_ror:1 (l <- l - 1)
__device__ __forceinline__ float dpp_partner(const float x) {
    int v = __float_as_int(x);
    v = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    v = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, true);
    return __int_as_float(v);
}

// Compile-time slot profile of the sparse filterbank fast path: group i holds the mels 16 i .. 16 i + 15 (one per lane)
// and each of them has at most kSlots[i] non-zero weights (default NeMo bank: 2 2 2 3 4 6 9 13).  Banks that do not fit
// (other n_fft / sample rates / > 128 mels) take the generic loop.
constexpr int kFastGroups = 8;
__host__ __device__ constexpr int fast_slots(int i) { return i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 2 : i == 3 ? 3 : i == 4 ? 4 : i == 5 ? 6 : i == 6 ? 9 : 13; }
__host__ __device__ constexpr int fast_slot_base(int i) { int o = 0; for (int k = 0; k < i; ++k) o += fast_slots(k); return o; }
constexpr int kFastSlots = fast_slot_base(kFastGroups);  // 41

constexpr int kRegionFloatsPk = 2 * 16 * kEStride;  // packed kernel: 16 x 17 frame pairs, 64-bit accesses only (one 16-lane group per LDS cycle)
constexpr int kPkHop = 160;  // hop of the frame-pair packed kernel (compile-time: the two frames of a lane are read by one ds_read2_b32)

constexpr int kStageVec = 6;  // float4 loads per thread and tile held in registers while the previous tile is computed

struct TileInfo {
    const float *x;  // utterance samples
    float *ob;       // utterance output
    int64_t len, n0;
    float lastv;
    int t0, T;
    bool stage;      // tile has frames to compute
    bool interior;   // the whole staged span (and the sample before it) lies inside the utterance
};

// global filterbank slot s -> (mel group i, slot j inside the group), resolved at compile time inside unrolled loops
template <class F>
__device__ __forceinline__ void constexpr_for_slot(const int s, F &&fn) {
#pragma unroll
    for (int i = 0; i < kFastGroups; ++i)
        if (s >= fast_slot_base(i) && s < fast_slot_base(i) + fast_slots(i)) fn(i, s - fast_slot_base(i));
}

// Both frames of a 16-lane group straight from the staged samples: v.re[n1] = (x[32 n1], x[32 n1 + hop]) and v.im[n1] the same
// one sample later, x = the lane's pointer (frame start + 2 lane).  One ds_read2_b32 per register pair.  Written in assembly
// because the load combiner pairs LDS reads by ascending offset — (x[32 n1], x[32 n1 + 32]) — which costs four register moves
// per pair to re-pair by frame; the wait for the 32 reads is part of the block (the compiler does not track them).
template <bool EZ>
__device__ __forceinline__ void load_frame_pair(const float *x, fa::melpk::LanePk &v) {
    static_assert(kPkHop == 160, "offsets below are written for hop 160");
    const unsigned a0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const float *)x));
#define FA_RD(i, b, o0, o1) "ds_read2_b32 %" #i ", %" #b " offset0:" #o0 " offset1:" #o1 "\n"
#define FA_RD4(i0, i1, i2, i3, b) FA_RD(i0, b, 0, 160) FA_RD(i1, b, 1, 161) FA_RD(i2, b, 32, 192) FA_RD(i3, b, 33, 193)
    if (EZ) {   // blocks n1 = 0 and n1 = 15 meet a zero window: literal zeros, 28 reads
        const fa::melpk::f2 zero = {0.0f, 0.0f};
        v.re[0] = zero; v.im[0] = zero; v.re[15] = zero; v.im[15] = zero;
        asm volatile(FA_RD(0, 28, 32, 192) FA_RD(1, 28, 33, 193) FA_RD4(2, 3, 4, 5, 29) FA_RD4(6, 7, 8, 9, 30) FA_RD4(10, 11, 12, 13, 31)
                     FA_RD4(14, 15, 16, 17, 32) FA_RD4(18, 19, 20, 21, 33) FA_RD4(22, 23, 24, 25, 34) FA_RD(26, 35, 0, 160) FA_RD(27, 35, 1, 161)
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(v.re[1]), "=&v"(v.im[1]), "=&v"(v.re[2]), "=&v"(v.im[2]), "=&v"(v.re[3]), "=&v"(v.im[3]), "=&v"(v.re[4]), "=&v"(v.im[4]),
                       "=&v"(v.re[5]), "=&v"(v.im[5]), "=&v"(v.re[6]), "=&v"(v.im[6]), "=&v"(v.re[7]), "=&v"(v.im[7]), "=&v"(v.re[8]), "=&v"(v.im[8]),
                       "=&v"(v.re[9]), "=&v"(v.im[9]), "=&v"(v.re[10]), "=&v"(v.im[10]), "=&v"(v.re[11]), "=&v"(v.im[11]), "=&v"(v.re[12]), "=&v"(v.im[12]),
                       "=&v"(v.re[13]), "=&v"(v.im[13]), "=&v"(v.re[14]), "=&v"(v.im[14])
                     : "v"(a0), "v"(a0 + 256), "v"(a0 + 512), "v"(a0 + 768), "v"(a0 + 1024), "v"(a0 + 1280), "v"(a0 + 1536), "v"(a0 + 1792)
                     : "memory");
    } else {
        asm volatile(FA_RD4(0, 1, 2, 3, 32) FA_RD4(4, 5, 6, 7, 33) FA_RD4(8, 9, 10, 11, 34) FA_RD4(12, 13, 14, 15, 35)
                     FA_RD4(16, 17, 18, 19, 36) FA_RD4(20, 21, 22, 23, 37) FA_RD4(24, 25, 26, 27, 38) FA_RD4(28, 29, 30, 31, 39)
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(v.re[0]), "=&v"(v.im[0]), "=&v"(v.re[1]), "=&v"(v.im[1]), "=&v"(v.re[2]), "=&v"(v.im[2]), "=&v"(v.re[3]), "=&v"(v.im[3]),
                       "=&v"(v.re[4]), "=&v"(v.im[4]), "=&v"(v.re[5]), "=&v"(v.im[5]), "=&v"(v.re[6]), "=&v"(v.im[6]), "=&v"(v.re[7]), "=&v"(v.im[7]),
                       "=&v"(v.re[8]), "=&v"(v.im[8]), "=&v"(v.re[9]), "=&v"(v.im[9]), "=&v"(v.re[10]), "=&v"(v.im[10]), "=&v"(v.re[11]), "=&v"(v.im[11]),
                       "=&v"(v.re[12]), "=&v"(v.im[12]), "=&v"(v.re[13]), "=&v"(v.im[13]), "=&v"(v.re[14]), "=&v"(v.im[14]), "=&v"(v.re[15]), "=&v"(v.im[15])
                     : "v"(a0), "v"(a0 + 256), "v"(a0 + 512), "v"(a0 + 768), "v"(a0 + 1024), "v"(a0 + 1280), "v"(a0 + 1536), "v"(a0 + 1792)
                     : "memory");
    }
#undef FA_RD4
#undef FA_RD
}

__device__ __forceinline__ void set_prio(const int p) {   // s_setprio takes an immediate
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// PK: frame-pair packed arithmetic (mel_pk.h): one pass of 2 frames per 16-lane group instead of two passes of one; needs
// FAST and hop == kPkHop.  PK == 2: additionally the window is zero on the outer 32 positions of the frame (fft256<EZ>).
template <int LAYOUT, bool FAST, int PK>
__global__ __launch_bounds__(kThreads, 2) void mel_kernel(const MelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *samples = smem;
    float *regions = samples + a.stage_alloc;
    float *outs = regions + kRegions * (PK ? kRegionFloatsPk : kRegionFloats);
    int32_t *mtab = reinterpret_cast<int32_t *>(outs + a.out_alloc);
    float *mw = reinterpret_cast<float *>(mtab + kMaxMels);
    float *wtab = mw + ((a.n_weights + 24 + 3) & ~3);   // packed kernel: per-lane window rows (mel_pk.h)

    const int tid = threadIdx.x;
    const int l = tid & (kGroup - 1);   // lane inside the 16-lane frame group
    const int grp = tid >> 4;           // 0..15: LDS region == (wave, frame of the pass)

    for (int i = tid; i < a.n_mels; i += kThreads) mtab[i] = a.mel_tab[i];
    for (int i = tid; i < a.n_weights; i += kThreads) mw[i] = PK ? 0.25f * a.mel_w[i] : a.mel_w[i];  // PK power bins carry an exact factor 4

    LaneConst kc;
    fa::melpk::LaneConstPk kp;
    if (PK) {
        fa::melpk::lane_const_init(l, a.tw256, a.tw512, kp);
        fa::melpk::window_table_fill(tid, kThreads, a.windowz, wtab);
    } else {
        Tables c;
        c.windowz = a.windowz;
        c.tw256 = reinterpret_cast<const float *>(a.tw256);
        c.tw512 = reinterpret_cast<const float *>(a.tw512);
        lane_const_init(l, c, kc);
    }
    int mlo[(kFastGroups + 2) / 3];  // first power bin of the lane's mel in every group, 3 x 10 bits per register (fast path)
#pragma unroll
    for (int i = 0; i < (kFastGroups + 2) / 3; ++i) mlo[i] = 0;
#pragma unroll
    for (int i = 0; i < kFastGroups; ++i)
        if (FAST && l + 16 * i < a.n_mels) mlo[i / 3] |= (a.mel_tab[l + 16 * i] & 1023) << (10 * (i % 3));
    __syncthreads();

    const int64_t per = (a.total_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * per;
    const int64_t stop = first + per < a.total_tiles ? first + per : a.total_tiles;
    const int n_mels = a.n_mels;
    const bool hop_even = (a.hop & 1) == 0;
    const bool vec_stage = a.stage_alloc <= kStageVec * kThreads * 4;

    auto tile_info = [&](const int64_t tl) {
        TileInfo ti;
        // workgroup-uniform values, forced into SGPRs: the metadata loads below become scalar loads
        const int tli = __builtin_amdgcn_readfirstlane(static_cast<int>(tl));
        const int b = tli / a.tiles_per_utt;
        ti.t0 = (tli - b * a.tiles_per_utt) * kTileFrames;
        // The plan's tables are read-only for the kernel: read them through the constant address space so that they become
        // scalar loads (lgkmcnt).  As plain global loads they are vector loads, and the s_waitcnt vmcnt(0) in front of their
        // first use also sits out every output store and prefetch load still in flight — about 2 k cycles per tile.
        typedef const int32_t __attribute__((address_space(4))) *c_i32;
        typedef const int64_t __attribute__((address_space(4))) *c_i64;
        typedef const float __attribute__((address_space(4))) *c_f32;
        ti.T = ((c_i32)a.frames)[b];
        const int64_t base = ((c_i64)a.offsets)[b];
        ti.len = ((c_i64)a.offsets)[b + 1] - base;
        ti.x = a.pcm + base;
        ti.ob = a.out + static_cast<int64_t>(b) * a.utt_stride;
        ti.lastv = a.last ? ((c_f32)a.last)[b] : 0.0f;
        ti.n0 = static_cast<int64_t>(ti.t0) * a.hop - a.pad;
        ti.stage = ti.t0 < ti.T;
        ti.interior = ti.n0 >= 1 && ti.n0 + kStageVec * kThreads * 4 <= ti.len;
        if (ti.t0 == 0 && tid == 0 && a.lengths) a.lengths[b] = ti.T;
        return ti;
    };
    // raw samples of a tile: thread t owns the float4 groups t, t + 256, ... of the staged span, plus the sample before each
    float4 raw[kStageVec];
    float rprev[kStageVec];
    auto fetch = [&](const TileInfo &ti) {
        if (ti.interior) {  // workgroup-uniform: straight-line independent loads, one memory round trip for the whole tile
#pragma unroll
            for (int r = 0; r < kStageVec; ++r) {
                const float *src = ti.x + ti.n0 + 4 * (tid + kThreads * r);
                raw[r] = *reinterpret_cast<const float4 *>(src);
                // the sample before the group through an opaque index: seen together, the two loads are re-cut into
                // (x[-1..2], x[3]) and the pieces copied into place right behind the loads, i.e. behind an s_waitcnt vmcnt
                int before = -1;
                asm volatile("" : "+v"(before));
                rprev[r] = src[before];
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < kStageVec; ++r) {  // utterance edges (first / last tiles): unconditional loads from clamped indices
            const int64_t n = ti.n0 + 4 * (tid + kThreads * r);
            float t[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {  // values outside [0, len) are replaced in stage(): no select (= no wait) behind the loads here
                const int64_t m = n - 1 + c;
                t[c] = ti.x[m < 0 ? 0 : (m < ti.len ? m : ti.len - 1)];
            }
            rprev[r] = t[0];
            raw[r] = make_float4(t[1], t[2], t[3], t[4]);
        }
    };
    auto stage = [&](const TileInfo &ti) {  // pre-emphasis (:211,:219-225: y[n] = x[n] - p x[n-1]); zero outside [0, len)
        int ts = tid;
        asm volatile("" : "+v"(ts));  // distinct from fetch()'s index arithmetic: shared, it would stay live (and spill) across the pass
        // Pin the prefetched values to this point.  The pre-emphasis below is vectorised into packed fma's whose operand pairs
        // (x[n - 1], x[n]) are assembled by register copies; without the pin those copies are scheduled right behind the loads
        // in fetch() — a whole pass earlier — together with the s_waitcnt vmcnt that makes the loads synchronous.
#pragma unroll
        for (int r = 0; r < kStageVec; ++r) asm volatile("" : "+v"(raw[r].x), "+v"(raw[r].y), "+v"(raw[r].z), "+v"(raw[r].w), "+v"(rprev[r]));
#pragma unroll
        for (int r = 0; r < kStageVec; ++r) {
            const int e = 4 * (ts + kThreads * r);
            if (e >= a.stage_alloc) continue;
            if (!ti.interior) {  // x[-1] = the carried last sample (:211), x = 0 elsewhere outside the utterance
                const int64_t n = ti.n0 + e;
                auto fix = [&](const float v, const int64_t m) { return m >= 0 && m < ti.len ? v : (m == -1 ? ti.lastv : 0.0f); };
                rprev[r] = fix(rprev[r], n - 1);
                raw[r].x = fix(raw[r].x, n); raw[r].y = fix(raw[r].y, n + 1); raw[r].z = fix(raw[r].z, n + 2); raw[r].w = fix(raw[r].w, n + 3);
            }
            float4 y;
            y.x = raw[r].x - a.preemph * rprev[r];
            y.y = raw[r].y - a.preemph * raw[r].x;
            y.z = raw[r].z - a.preemph * raw[r].y;
            y.w = raw[r].w - a.preemph * raw[r].z;
            if (!ti.interior) {
                const int64_t n = ti.n0 + e;
                if (n < 0 || n >= ti.len) y.x = 0.f;
                if (n + 1 < 0 || n + 1 >= ti.len) y.y = 0.f;
                if (n + 2 < 0 || n + 2 >= ti.len) y.z = 0.f;
                if (n + 3 < 0 || n + 3 >= ti.len) y.w = 0.f;
            }
            *reinterpret_cast<float4 *>(samples + e) = y;
        }
    };

    unsigned long long t_seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = clock64();
#define MEL_STAMP(i) do { if (a.prof) { const unsigned long long t_now = clock64(); t_seg[i] += t_now - t_prev; t_prev = t_now; } } while (0)
#ifdef FA_MEL_PROF_FINE   // diagnostics build: split the packed pass (slots 8..11; slot 3 keeps the remainder)
#define MEL_STAMP_FINE(i) MEL_STAMP(i)
#else
#define MEL_STAMP_FINE(i) do { } while (0)
#endif
    // All table / constant loads are complete from here on (vmcnt(0), expcnt/lgkmcnt untouched).  Without this explicit
    // instruction the wait-count pass keeps them "possibly pending" around the tile loop and guards the first LDS reads of
    // every pass with s_waitcnt vmcnt(<=3), which also drains the next tile's prefetch loads (vmcnt retires in order) and
    // exposes a full HBM round trip per tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    auto stage_tile = [&](const TileInfo &ti) {
        if (!ti.stage) return;
        if (vec_stage) stage(ti);
        else
            for (int i = tid; i < a.stage_count; i += kThreads) {  // large hops: plain staging loop
                const int64_t n = ti.n0 + i;
                float y = 0.0f;
                if (n >= 0 && n < ti.len) y = ti.x[n] - a.preemph * (n > 0 ? ti.x[n - 1] : ti.lastv);
                samples[i] = y;
            }
    };
    auto info_and_fetch = [&](const int64_t tl) {   // metadata + the 12 loads of tile tl (a no-op tile past the range)
        TileInfo ti{};
        ti.stage = false;
        if (tl < stop) { ti = tile_info(tl); if (ti.stage && vec_stage) fetch(ti); }
        return ti;
    };
    TileInfo cur = info_and_fetch(first);
    stage_tile(cur);
    TileInfo nxt = info_and_fetch(first + 1);   // always one tile ahead: its samples sit in registers during the pass

    // Per tile: barrier | compute | barrier | stage the next tile | store this tile | issue the loads of the tile after.  Staging
    // BEFORE the stores matters: the wait in front of the staging is vmcnt(0) (the wait-count pass cannot bound it across the
    // loop), and with the four output stores issued first it would sit out their write acknowledgements (~1.5 k cycles per
    // tile); in this order the only VMEM operations in flight are the loads issued a whole pass earlier.
    for (int64_t tl = first; tl < stop; ++tl) {
        __syncthreads();
        MEL_STAMP(1);
        if (PK && cur.stage) {  // workgroup-uniform; group g computes the frames 2 g and 2 g + 1 of the tile in one pass
            using namespace fa::melpk;
            const int f = 2 * grp;
            LanePk v;
            float4 w4[8];
            set_prio(a.prio_rd);
#pragma unroll
            for (int q = 0; q < 8; ++q) w4[q] = reinterpret_cast<const float4 *>(wtab)[q * kGroup + l];
            load_frame_pair<PK == 2>(samples + f * kPkHop + 2 * l, v);
            MEL_STAMP_FINE(8);
            f2 *P2 = reinterpret_cast<f2 *>(__builtin_assume_aligned(regions + grp * kRegionFloatsPk, 8));
            set_prio(a.prio_lo);   // VALU-bound stretch: yield issue slots to the co-resident wave's short latency-bound bursts
            fft256<PK == 2>(l, v, w4, kp, P2);
            MEL_STAMP_FINE(9);
            set_prio(a.prio_pw);
            // power bins of both frames, pair k at P2[k], over the transpose buffer (its last reads are already issued)
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // Z[256 - (l + 16 j)]: lane (16 - l) & 15, register 15 - j (lane 0: own (16 - j) & 15)
                const f2 pr = partner(v.re[15 - j]), pi = partner(v.im[15 - j]);
                const f2 qr = l == 0 ? v.re[(16 - j) & 15] : pr;
                const f2 qi = l == 0 ? v.im[(16 - j) & 15] : pi;
                f2 plo, phi;
                pair_power4(v.re[j], v.im[j], qr, qi, kp.t2[j], plo, phi);
                P2[l + 16 * j] = plo;
                P2[kHalf - (l + 16 * j)] = phi;
            }
            if (l == 0) P2[128] = 4.0f * (v.re[8] * v.re[8] + v.im[8] * v.im[8]);  // k = 128: X = conj(Z[128])
            MEL_STAMP_FINE(10);
            set_prio(a.prio_hi);   // LDS-latency-bound from here to the next pass: issue as soon as data arrives
            // sparse triangular filterbank (vDSP_mmul row, :270-283, zeros skipped): lane l owns the mels l + 16 i; weights are
            // fetched two global slots (2 p, 2 p + 1) per register pair, so a pair may straddle two mel groups
            f2 acc[kFastGroups];
            const f2 *P[kFastGroups];
#pragma unroll
            for (int i = 0; i < kFastGroups; ++i) {
                acc[i] = f2{0.0f, 0.0f};
                P[i] = P2 + ((mlo[i / 3] >> (10 * (i % 3))) & 1023);
            }
            // All LDS reads of a half are issued before its first fma (sched_barrier keeps the scheduler from re-serialising
            // them into load -> wait -> fma chains, which exposes one LDS round trip per slot): weights first, then the bins.
            constexpr int kSplit = fast_slot_base(6) & ~1;   // slots [0, kSplit): groups 0..5 (even, so weight pairs do not straddle the halves)
            {
                f2 w[kSplit / 2], pb[kSplit];
#pragma unroll
                for (int sp = 0; sp < kSplit / 2; ++sp) w[sp] = f2{mw[(2 * sp) * kGroup + l], mw[(2 * sp + 1) * kGroup + l]};
#pragma unroll
                for (int s = 0; s < kSplit; ++s) constexpr_for_slot(s, [&](const int i, const int j) { pb[s] = P[i][j]; });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < kSplit; ++s)
                    constexpr_for_slot(s, [&](const int i, const int) { acc[i] = (s & 1) ? fma_hi(pb[s], w[s / 2], acc[i]) : fma_lo(pb[s], w[s / 2], acc[i]); });
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                constexpr int kRest = kFastSlots - kSplit;
                f2 w[(kRest + 1) / 2], pb[kRest];
#pragma unroll
                for (int sp = 0; sp < (kRest + 1) / 2; ++sp) w[sp] = f2{mw[(kSplit + 2 * sp) * kGroup + l], mw[(kSplit + 2 * sp + 1) * kGroup + l]};
#pragma unroll
                for (int s = kSplit; s < kFastSlots; ++s) constexpr_for_slot(s, [&](const int i, const int j) { pb[s - kSplit] = P[i][j]; });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = kSplit; s < kFastSlots; ++s)
                    constexpr_for_slot(s, [&](const int i, const int) {
                        acc[i] = (s & 1) ? fma_hi(pb[s - kSplit], w[(s - kSplit) / 2], acc[i]) : fma_lo(pb[s - kSplit], w[(s - kSplit) / 2], acc[i]);
                    });
            }
            // :542-549: log(acc + floor) or log(max(acc, floor)) as log(max(acc + add, clamp)) with wave-uniform add / clamp.
            // The argument is >= floor > 0 and far from the denormal range, so the hardware log2 (1 ulp) times ln 2 replaces
            // logf's denormal pre-scaling and two-term ln 2 product: 2 instructions instead of 11.
            MEL_STAMP_FINE(11);
            const float add_floor = a.floor_clamped ? 0.0f : a.log_floor, clamp_floor = a.floor_clamped ? a.log_floor : 0.0f;
#pragma unroll
            for (int i = 0; i < kFastGroups; ++i) {
                const int m = l + 16 * i;
                constexpr float kLn2 = 0.693147180559945309f;
                const f2 x = acc[i] + add_floor;
                f2 val;
                val.x = kLn2 * __builtin_amdgcn_logf(fmaxf(x.x, clamp_floor));
                val.y = kLn2 * __builtin_amdgcn_logf(fmaxf(x.y, clamp_floor));
                if (m < n_mels) {
                    if (LAYOUT == FA_MEL_LAYOUT_MEL_MAJOR) *reinterpret_cast<f2 *>(outs + m * kMelPad + f) = val;
                    else { outs[f * (n_mels + kFramePad) + m] = val.x; outs[(f + 1) * (n_mels + kFramePad) + m] = val.y; }
                }
            }
        }
        if (!PK && cur.stage) {  // workgroup-uniform
            float *R = static_cast<float *>(__builtin_assume_aligned(regions + grp * kRegionFloats, 8));
#pragma unroll 1
            for (int pass = 0; pass < kPasses; ++pass) {
                // frame of this 16-lane group inside the tile: wave w owns frames [8w, 8w + 8)
                const int f = (grp >> 2) * (kWaveFrames * kPasses) + pass * kWaveFrames + (grp & 3);
                const float *fs = hop_even ? static_cast<const float *>(__builtin_assume_aligned(samples + f * a.hop, 8)) : samples + f * a.hop;
                Lane v;
                if (hop_even) {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const float2 s2 = *reinterpret_cast<const float2 *>(fs + 32 * n1 + 2 * l);
                        v.re[n1] = s2.x; v.im[n1] = s2.y;
                    }
                } else {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) { v.re[n1] = fs[32 * n1 + 2 * l]; v.im[n1] = fs[32 * n1 + 2 * l + 1]; }
                }
                // Everything below exchanges data only between the 16 lanes of one frame group, i.e. inside one
                // wavefront: LDS operations of a wavefront complete in program order, no barrier is needed.
                phase_a2(l, v, kc, R);
                phase_b1(l, R, v);
                float qr[8], qi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // Z[256 - (l + 16 j)]: lane (16 - l) & 15, register 15 - j (lane 0: own (16 - j) & 15)
                    const float pr = dpp_partner(v.re[15 - j]), pi = dpp_partner(v.im[15 - j]);
                    qr[j] = l == 0 ? v.re[(16 - j) & 15] : pr;
                    qi[j] = l == 0 ? v.im[(16 - j) & 15] : pi;
                }
                Power p;
                phase_c1v2(l, v, qr, qi, kc, p);
                // power bins at R + {0,14,28,10}[frame]: regions are 578 = 2 (mod 32) floats apart, so this puts the two frames
                // of a 32-lane LDS service group 16 banks apart and the gathers below stop colliding
                float *PR = R + ((0x0a1c0e00u >> (8 * (grp & 3))) & 0xff);
                phase_c2(l, p, PR);
                // sparse triangular filterbank + log for this group's frame: lane l owns mels l, l + 16, ...
                if (FAST) {
                    // weights zero-padded to the slot profile, [slot][16 lanes]: every read below is independent
                    float acc[kFastGroups];
#pragma unroll
                    for (int i = 0; i < kFastGroups; ++i) {
                        acc[i] = 0.0f;
#pragma unroll
                        for (int j = 0; j < fast_slots(i); ++j)
                            acc[i] += mw[(fast_slot_base(i) + j) * kGroup + l] * PR[((mlo[i / 3] >> (10 * (i % 3))) & 1023) + j];  // vDSP_mmul row (:270-283), zeros skipped
                    }
#pragma unroll
                    for (int i = 0; i < kFastGroups; ++i) {
                        const int m = l + 16 * i;
                        const float val = a.floor_clamped ? logf(fmaxf(acc[i], a.log_floor)) : logf(acc[i] + a.log_floor);  // :542-549
                        if (m < n_mels) {
                            if (LAYOUT == FA_MEL_LAYOUT_MEL_MAJOR) outs[m * kMelPad + f] = val;
                            else outs[f * (n_mels + kFramePad) + m] = val;
                        }
                    }
                } else {
                    for (int m = l; m < n_mels; m += kGroup) {
                        const int packed = mtab[m];
                        const int lo = packed & 1023, cnt = (packed >> 10) & 1023, st = packed >> 20;
                        const float *P = PR + lo;
                        const float *wgt = mw + st;
                        float acc = 0.0f;
                        for (int j = 0; j < cnt; ++j) acc += wgt[j] * P[j];
                        const float val = a.floor_clamped ? logf(fmaxf(acc, a.log_floor)) : logf(acc + a.log_floor);
                        if (LAYOUT == FA_MEL_LAYOUT_MEL_MAJOR) outs[m * kMelPad + f] = val;
                        else outs[f * (n_mels + kFramePad) + m] = val;
                    }
                }
            }
        }
        MEL_STAMP(3);
        __syncthreads();
        MEL_STAMP(4);
        stage_tile(nxt);   // `samples` has no reader left after the barrier
        MEL_STAMP(0);

        const bool full_tile = cur.t0 + kTileFrames <= cur.T && cur.t0 + kTileFrames <= a.frame_stride;
        if (LAYOUT == FA_MEL_LAYOUT_MEL_MAJOR && FAST && full_tile && n_mels == kFastGroups * kGroup) {
            // 8 threads per mel row, 4 consecutive frames each: 16-byte LDS reads, 16-byte global stores (:287); a fixed number
            // of stores per thread keeps the wait count for the next tile's prefetch exact (vmcnt(4) instead of vmcnt(0))
            int st = tid;
            asm volatile("" : "+v"(st));   // keep the store addresses loop-variant: hoisted, they sit in 12 registers across the whole tile loop
#pragma unroll
            for (int it = 0; it < kFastGroups * kGroup * (kTileFrames / 4) / kThreads; ++it) {
                const int idx = st + kThreads * it, m = idx >> 3, f = (idx & 7) * 4;   // every thread stores: no exec masking, exact vmcnt
                *reinterpret_cast<float4 *>(cur.ob + static_cast<int64_t>(m) * a.frame_stride + cur.t0 + f) =
                    *reinterpret_cast<const float4 *>(outs + m * kMelPad + f);
            }
        } else if (LAYOUT == FA_MEL_LAYOUT_MEL_MAJOR) {
            for (int idx = tid; idx < n_mels * (kTileFrames / 4); idx += kThreads) {
                const int m = idx >> 3, f = (idx & 7) * 4, t = cur.t0 + f;
                if (t >= a.frame_stride) continue;
                float4 v4 = *reinterpret_cast<const float4 *>(outs + m * kMelPad + f);
                float *dst = cur.ob + static_cast<int64_t>(m) * a.frame_stride + t;
                if (t + 3 < cur.T && t + 3 < a.frame_stride) { *reinterpret_cast<float4 *>(dst) = v4; continue; }
                const float e[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (t + c < a.frame_stride) dst[c] = t + c < cur.T ? e[c] : 0.0f;  // padValue (:39) for t >= T
            }
        } else {
            const int work = kTileFrames * n_mels;
            for (int idx = tid; idx < work; idx += kThreads) {
                const int f = idx / n_mels, m = idx - f * n_mels;
                const int t = cur.t0 + f;
                if (t >= a.frame_stride) continue;
                cur.ob[static_cast<int64_t>(t) * n_mels + m] = t < cur.T ? outs[f * (n_mels + kFramePad) + m] : 0.0f;  // :451
            }
        }
        // the barrier at the top of the next iteration orders these `outs` reads before the next tile's writes
        MEL_STAMP(5);
        cur = nxt;
        // the loads of the tile after the next travel from HBM during the barrier wait and the whole next pass
        nxt = info_and_fetch(tl + 2);
        MEL_STAMP(2);
    }
    if (a.prof && blockIdx.x == gridDim.x / 2 && tid == 0) {
        for (int i = 0; i < 12; ++i) if (i != 7) atomicAdd(&a.prof[i], t_seg[i]);
        atomicAdd(&a.prof[7], static_cast<unsigned long long>(stop - first));
    }
#undef MEL_STAMP
}

#include "mel_v4.inc"

// NeMo per_feature normalisation as done by UnifiedMelExtractor.normalizePerFeature
// (reference: Sources/FluidAudio/ASR/Parakeet/Unified/UnifiedMelExtractor.swift:91-113): for every mel bin subtract the
// mean and divide by the unbiased std (+1e-5) over the valid frames; frames >= valid become 0; valid == 0 zeroes the row.
// One wavefront per (utterance, mel) row of a [B][n_mels][frame_stride] tensor; rows of up to 2048 frames are held in
// registers between the three passes (sum, centred squares, write): one HBM read and one write per element.
constexpr int kNormRegs = 32;   // frames per lane held in registers: rows up to 2048 frames are read from HBM once

__global__ __launch_bounds__(256) void mel_norm_kernel(float *__restrict__ mel, const int32_t *__restrict__ valid_frames, int64_t rows,
                                                         int32_t n_mels, int32_t frame_stride, int32_t frames) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int b = static_cast<int>(row / n_mels);
    int valid = valid_frames[b];
    valid = valid < 0 ? 0 : (valid > frames ? frames : valid);
    float *x = mel + row * frame_stride;
    const bool in_regs = frames <= 64 * kNormRegs;
    float v[kNormRegs];
    float mean = 0.0f, inv_std = 0.0f;
    if (valid > 0) {
        float sum = 0.0f;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < kNormRegs; ++j) { const int t = lane + 64 * j; v[j] = t < valid ? x[t] : 0.0f; sum += v[j]; }
        } else {
            for (int t = lane; t < valid; t += 64) sum += x[t];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        mean = sum / static_cast<float>(valid);
        float var = 0.0f;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < kNormRegs; ++j) if (lane + 64 * j < valid) { const float dlt = v[j] - mean; var += dlt * dlt; }
        } else {
            for (int t = lane; t < valid; t += 64) { const float dlt = x[t] - mean; var += dlt * dlt; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off);
        const float denom = static_cast<float>(valid > 1 ? valid - 1 : 1);
        inv_std = 1.0f / (sqrtf(var / denom) + 1e-5f);
    }
    if (in_regs && valid > 0) {
#pragma unroll
        for (int j = 0; j < kNormRegs; ++j) { const int t = lane + 64 * j; if (t < frames) x[t] = t < valid ? (v[j] - mean) * inv_std : 0.0f; }
    } else {
        for (int t = lane; t < frames; t += 64) x[t] = t < valid ? (x[t] - mean) * inv_std : 0.0f;
    }
}

// ----------------------------------------------------------------------------- host tables
// createHannWindow (:553-562)
void make_hann(int win, bool periodic, std::vector<float> &w) {
    w.resize(win);
    const float divisor = periodic ? static_cast<float>(win) : static_cast<float>(win - 1);
    const float pi_f = static_cast<float>(M_PI);
    for (int i = 0; i < win; ++i) {
        const float phase = 2.0f * pi_f * static_cast<float>(i) / divisor;
        w[i] = 0.5f * (1.0f - cosf(phase));
    }
}

float hz_to_mel(float hz) {  // :575-586
    const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp, log_step = logf(6.4f) / 27.0f;
    return hz >= min_log_hz ? min_log_mel + logf(hz / min_log_hz) / log_step : hz / f_sp;
}
float mel_to_hz(float mel) {  // :588-599
    const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp, log_step = logf(6.4f) / 27.0f;
    return mel >= min_log_mel ? min_log_hz * expf(log_step * (mel - min_log_mel)) : f_sp * mel;
}

// createMelFilterbank (:564-642), dense [n_mels][bins]
void make_filterbank(int n_fft, int n_mels, int sr, std::vector<float> &fb) {
    const int bins = n_fft / 2 + 1;
    fb.assign(static_cast<size_t>(n_mels) * bins, 0.0f);
    const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(static_cast<float>(sr) / 2.0f);
    std::vector<float> pts(n_mels + 2), freqs(bins);
    for (int i = 0; i < n_mels + 2; ++i)
        pts[i] = mel_to_hz(mel_min + static_cast<float>(i) * (mel_max - mel_min) / static_cast<float>(n_mels + 1));
    for (int i = 0; i < bins; ++i) freqs[i] = static_cast<float>(i) * static_cast<float>(sr) / static_cast<float>(n_fft);
    for (int m = 0; m < n_mels; ++m) {
        const float fl = pts[m], fc = pts[m + 1], fr = pts[m + 2];
        const float norm = 2.0f / (fr - fl);
        for (int k = 0; k < bins; ++k) {
            const float f = freqs[k];
            if (f >= fl && f < fc) fb[static_cast<size_t>(m) * bins + k] = norm * (f - fl) / (fc - fl);
            else if (f >= fc && f <= fr) fb[static_cast<size_t>(m) * bins + k] = norm * (fr - f) / (fr - fc);
        }
    }
}

// torchaudio melscale_fbanks(norm: nil, mel_scale: "htk") as built by LuxTtsMelExtractor.htkMelFilterbank
// (Sources/FluidAudio/TTS/LuxTts/LuxTtsMelExtractor.swift:160-189): double arithmetic, rounded to float at the end
void make_filterbank_htk(int n_fft, int n_mels, int sr, std::vector<float> &fb) {
    const int bins = n_fft / 2 + 1;
    fb.assign(static_cast<size_t>(n_mels) * bins, 0.0f);
    const double f_max = static_cast<double>(sr) / 2.0;
    auto hz_to_mel_htk = [](double hz) { return 2595.0 * log10(1.0 + hz / 700.0); };
    auto mel_to_hz_htk = [](double mel) { return 700.0 * (pow(10.0, mel / 2595.0) - 1.0); };
    const double mel_min = hz_to_mel_htk(0.0), mel_max = hz_to_mel_htk(f_max);
    std::vector<double> pts(n_mels + 2), freqs(bins);
    for (int i = 0; i < n_mels + 2; ++i) pts[i] = mel_to_hz_htk(mel_min + static_cast<double>(i) * (mel_max - mel_min) / static_cast<double>(n_mels + 1));
    for (int b = 0; b < bins; ++b) freqs[b] = static_cast<double>(b) * f_max / static_cast<double>(bins - 1);
    for (int m = 0; m < n_mels; ++m)
        for (int b = 0; b < bins; ++b) {
            const double up = (freqs[b] - pts[m]) / (pts[m + 1] - pts[m]), down = (pts[m + 2] - freqs[b]) / (pts[m + 2] - pts[m + 1]);
            const double v = up < down ? up : down;
            fb[static_cast<size_t>(m) * bins + b] = static_cast<float>(v > 0.0 ? v : 0.0);
        }
}

// the bank a configuration asks for: caller's table, HTK/no-norm, or the reference's Slaney bank
void config_filterbank(const fa_mel_config *c, std::vector<float> &fb) {
    const size_t n = static_cast<size_t>(c->n_mels) * (c->n_fft / 2 + 1);
    if (c->filterbank) fb.assign(c->filterbank, c->filterbank + n);
    else if (c->mel_scale == FA_MEL_SCALE_HTK_NONORM) make_filterbank_htk(c->n_fft, c->n_mels, c->sample_rate, fb);
    else make_filterbank(c->n_fft, c->n_mels, c->sample_rate, fb);
}

fa_status validate(const fa_mel_config *c) {
    if (!c) return FA_INVALID_ARGUMENT;
    if (c->n_fft < 64 || c->n_fft > 2048 || (c->n_fft & (c->n_fft - 1)) != 0) return FA_INVALID_ARGUMENT;  // power of two
    if (c->win < 2 || c->win > c->n_fft || c->hop < 1 || c->hop > 4096) return FA_INVALID_ARGUMENT;
    if (c->n_mels < 1 || c->n_mels > kMaxMels || c->sample_rate < 1) return FA_INVALID_ARGUMENT;
    if (c->padding_mode < 0 || c->padding_mode > 2 || c->layout < 0 || c->layout > 1) return FA_INVALID_ARGUMENT;
    if (c->floor_mode < 0 || c->floor_mode > 1) return FA_INVALID_ARGUMENT;
    if (c->power != 0.0f && c->power != 1.0f && c->power != 2.0f) return FA_INVALID_ARGUMENT;
    if (c->center_pad < 0 || c->center_pad > 1 || c->mel_scale < 0 || c->mel_scale > 1 || c->tail_mode < 0 || c->tail_mode > 1) return FA_INVALID_ARGUMENT;
    return FA_SUCCESS;
}

// configurations the tuned n_fft = 512 kernels do not cover take mel_generic_kernel
bool needs_generic(const fa_mel_config *c) {
    return c->n_fft != kNfft || c->power == 1.0f || (c->center_pad == FA_MEL_CENTER_REFLECT && c->padding_mode == FA_MEL_PAD_CENTER) ||
           c->tail_mode == FA_MEL_TAIL_REPLICATE || fa::sw_on(fa::Sw::MEL_GENERIC);
}

}  // namespace

struct fa_mel_plan {
    fa_ctx *ctx = nullptr;
    fa_mel_config cfg{};
    int32_t batch = 0;
    int32_t frame_stride = 0;
    int64_t utt_stride = 0;
    int64_t total_frames = 0;
    int64_t total_samples = 0;
    void *dev = nullptr;  // one allocation holding every device-side table of the plan
    MelArgs args{};
    size_t lds_bytes = 0;
    int grid = 0;
    bool fast = false;  // filterbank fits the compile-time slot profile
    bool pk = false;    // frame-pair packed kernel (fast bank, hop == kPkHop)
    bool edge_zero = false;  // the zero-extended window vanishes on positions [0, 32) and [480, 512) of the frame
    int v4_wps = 0;          // > 0: mel_kernel_v4 with that many workgroups per CU (packed kernel, 128 mels, hop 160)
    unsigned long long launches = 0;   // v4 launches made so far (spaces the tile-queue ranges)
    bool v4_deep = true;               // tile queue two tiles ahead: the next tile's samples travel during the second pass (FA_MEL_V4_DEEP=0: one ahead)
    bool generic = false;    // mel_generic_kernel (any n_fft, magnitude, reflect padding, replicated tail)
    fa::melgen::GenArgs gargs{};
};

extern "C" {

void fa_mel_default_config(fa_mel_config *c) {
    if (!c) return;
    c->sample_rate = 16000; c->n_mels = 128; c->n_fft = 512; c->hop = 160; c->win = 400;
    c->preemph = 0.97f; c->pad_to = 0; c->log_floor = ldexpf(1.0f, -24);
    c->floor_mode = FA_MEL_FLOOR_ADDITIVE; c->window_periodic = 0;
    c->padding_mode = FA_MEL_PAD_CENTER; c->layout = FA_MEL_LAYOUT_MEL_MAJOR;
    c->power = 2.0f; c->center_pad = FA_MEL_CENTER_ZERO; c->mel_scale = FA_MEL_SCALE_SLANEY; c->tail_mode = FA_MEL_TAIL_ZERO;
    c->filterbank = nullptr;
}

int32_t fa_mel_num_frames(const fa_mel_config *c, int64_t n) {
    if (!c || n <= 0 || c->hop < 1) return 0;
    int64_t frames;
    switch (c->padding_mode) {
        case FA_MEL_PAD_CENTER: frames = 1 + (n + 2 * static_cast<int64_t>(c->n_fft / 2) - c->win) / c->hop; break;  // :195-197
        case FA_MEL_PAD_PREPADDED: frames = (n - c->n_fft) / c->hop + 1; if (frames < 0) frames = 0; break;          // :345
        default: frames = 1 + (n - c->win) / c->hop; break;                                                          // :133
    }
    if (frames <= 0) return 0;
    return frames > INT32_MAX ? 0 : static_cast<int32_t>(frames);
}

int32_t fa_mel_padded_frames(const fa_mel_config *c, int32_t frames) {
    if (!c) return 0;
    const int32_t p = c->pad_to > 1 ? c->pad_to : 1;  // :72
    return ((frames + p - 1) / p) * p;                // :204,:354
}

fa_status fa_mel_hann_window(const fa_mel_config *c, float *out) {
    if (!c || !out || c->win < 1) return FA_INVALID_ARGUMENT;
    std::vector<float> w;
    make_hann(c->win, c->window_periodic != 0, w);
    memcpy(out, w.data(), sizeof(float) * w.size());
    return FA_SUCCESS;
}

fa_status fa_mel_filterbank(const fa_mel_config *c, float *out) {
    if (!c || !out || c->n_fft < 2 || c->n_mels < 1) return FA_INVALID_ARGUMENT;
    std::vector<float> fb;
    config_filterbank(c, fb);
    memcpy(out, fb.data(), sizeof(float) * fb.size());
    return FA_SUCCESS;
}

fa_status fa_mel_plan_create(fa_ctx *ctx, const fa_mel_config *cfg, const int64_t *offsets, int32_t batch,
                             const int32_t *expected_frames, int32_t frame_stride, fa_mel_plan **out) {
    if (!ctx || !out) return FA_INVALID_ARGUMENT;
    *out = nullptr;
    if (validate(cfg) != FA_SUCCESS) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "mel: unsupported configuration");
    if (!offsets || batch < 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "mel: empty batch");
    fa::DeviceGuard guard(ctx->device);
    try {
        fa_mel_plan *p = new fa_mel_plan();
        p->ctx = ctx; p->cfg = *cfg; p->batch = batch;
        const int bins = cfg->n_fft / 2 + 1;
        std::vector<int32_t> frames(batch), natural(batch);
        int32_t max_padded = 1;
        for (int b = 0; b < batch; ++b) {
            const int64_t len = offsets[b + 1] - offsets[b];
            if (len < 0) { delete p; return fa::set_error(ctx, FA_INVALID_ARGUMENT, "mel: offsets not monotone"); }
            int32_t T = fa_mel_num_frames(cfg, len);
            natural[b] = T;
            if (expected_frames && len > 0) T = expected_frames[b] > 0 ? expected_frames[b] : 0;  // :347
            frames[b] = T;
            p->total_frames += T;
            const int32_t tp = T > 0 ? fa_mel_padded_frames(cfg, T) : 1;
            if (tp > max_padded) max_padded = tp;
        }
        p->total_samples = offsets[batch];
        if (frame_stride <= 0) frame_stride = max_padded;
        if (frame_stride < max_padded) { delete p; return fa::set_error(ctx, FA_OUTPUT_TOO_SMALL, "mel: frame_stride too small"); }
        p->frame_stride = frame_stride;
        p->utt_stride = static_cast<int64_t>(frame_stride) * cfg->n_mels;

        // tables
        std::vector<float> hann, fb;
        make_hann(cfg->win, cfg->window_periodic != 0, hann);
        config_filterbank(cfg, fb);
        const int off = cfg->padding_mode == FA_MEL_PAD_LEGACY ? 0 : (cfg->n_fft - cfg->win) / 2;  // :234 / :148-153
        if (needs_generic(cfg)) {
            // sparse rows (support lo..hi of every mel, interior zeros kept), window, twiddles exp(-2 pi i k / n_fft)
            const int N = cfg->n_fft;
            std::vector<int32_t> lo(cfg->n_mels, 0), cnt(cfg->n_mels, 0), start(cfg->n_mels, 0);
            std::vector<float> wts;
            for (int m = 0; m < cfg->n_mels; ++m) {
                int l0 = -1, h0 = -1;
                for (int k = 0; k < bins; ++k) if (fb[static_cast<size_t>(m) * bins + k] != 0.0f) { if (l0 < 0) l0 = k; h0 = k; }
                start[m] = static_cast<int32_t>(wts.size());
                if (l0 >= 0) { lo[m] = l0; cnt[m] = h0 - l0 + 1; for (int k = l0; k <= h0; ++k) wts.push_back(fb[static_cast<size_t>(m) * bins + k]); }
            }
            if (wts.empty()) wts.push_back(0.0f);
            std::vector<float2> tw(N / 2 + 1);
            for (int k = 0; k <= N / 2; ++k) { const double ang = -2.0 * M_PI * k / N; tw[k] = make_float2((float)cos(ang), (float)sin(ang)); }
            auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
            const size_t o_off = 0, o_fr = align(o_off + sizeof(int64_t) * (batch + 1)), o_nat = align(o_fr + sizeof(int32_t) * batch),
                         o_win = align(o_nat + sizeof(int32_t) * batch), o_tw = align(o_win + sizeof(float) * cfg->win), o_lo = align(o_tw + sizeof(float2) * tw.size()),
                         o_cnt = align(o_lo + sizeof(int32_t) * cfg->n_mels), o_st = align(o_cnt + sizeof(int32_t) * cfg->n_mels),
                         o_w = align(o_st + sizeof(int32_t) * cfg->n_mels), total = align(o_w + sizeof(float) * wts.size());
            std::vector<char> blob(total, 0);
            memcpy(blob.data() + o_off, offsets, sizeof(int64_t) * (batch + 1));
            memcpy(blob.data() + o_fr, frames.data(), sizeof(int32_t) * batch);
            memcpy(blob.data() + o_nat, natural.data(), sizeof(int32_t) * batch);
            memcpy(blob.data() + o_win, hann.data(), sizeof(float) * cfg->win);
            memcpy(blob.data() + o_tw, tw.data(), sizeof(float2) * tw.size());
            memcpy(blob.data() + o_lo, lo.data(), sizeof(int32_t) * cfg->n_mels);
            memcpy(blob.data() + o_cnt, cnt.data(), sizeof(int32_t) * cfg->n_mels);
            memcpy(blob.data() + o_st, start.data(), sizeof(int32_t) * cfg->n_mels);
            memcpy(blob.data() + o_w, wts.data(), sizeof(float) * wts.size());
            hipError_t e = hipMalloc(&p->dev, total);
            if (e != hipSuccess) { delete p; return fa::hip_status(ctx, e, "mel plan hipMalloc"); }
            e = hipMemcpy(p->dev, blob.data(), total, hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(p->dev); delete p; return fa::hip_status(ctx, e, "mel plan upload"); }
            char *d = static_cast<char *>(p->dev);
            fa::melgen::GenArgs &g = p->gargs;
            g.offsets = reinterpret_cast<const int64_t *>(d + o_off);
            g.frames = reinterpret_cast<const int32_t *>(d + o_fr);
            g.stft_frames = reinterpret_cast<const int32_t *>(d + o_nat);
            g.window = reinterpret_cast<const float *>(d + o_win);
            g.tw = reinterpret_cast<const float2 *>(d + o_tw);
            g.mel_lo = reinterpret_cast<const int32_t *>(d + o_lo);
            g.mel_cnt = reinterpret_cast<const int32_t *>(d + o_cnt);
            g.mel_start = reinterpret_cast<const int32_t *>(d + o_st);
            g.mel_w = reinterpret_cast<const float *>(d + o_w);
            g.utt_stride = p->utt_stride; g.batch = batch; g.frame_stride = frame_stride; g.n_mels = cfg->n_mels; g.n_fft = N;
            g.log2_m = 0; while ((2 << g.log2_m) < N) ++g.log2_m;           // log2(N / 2)
            g.win = cfg->win; g.off = off; g.hop = cfg->hop;
            g.pad = cfg->padding_mode == FA_MEL_PAD_CENTER ? N / 2 : 0;
            g.preemph = cfg->padding_mode == FA_MEL_PAD_LEGACY ? 0.0f : cfg->preemph;
            g.log_floor = cfg->log_floor; g.floor_clamped = cfg->floor_mode == FA_MEL_FLOOR_CLAMPED;
            g.reflect = cfg->center_pad == FA_MEL_CENTER_REFLECT && cfg->padding_mode == FA_MEL_PAD_CENTER;
            g.magnitude = cfg->power == 1.0f; g.tail_replicate = cfg->tail_mode == FA_MEL_TAIL_REPLICATE;
            g.frame_major = cfg->layout == FA_MEL_LAYOUT_FRAME_MAJOR;
            p->generic = true;
            p->lds_bytes = sizeof(float) * fa::melgen::kWaves * (2 * static_cast<size_t>(N) + 8);
            if (p->lds_bytes > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fa::melgen::mel_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(p->lds_bytes));
            const int64_t items = (static_cast<int64_t>(batch) * frame_stride + fa::melgen::kWaves - 1) / fa::melgen::kWaves;
            p->grid = static_cast<int>(items < 256 * 16 ? (items < 1 ? 1 : items) : 256 * 16);
            *out = p;
            return FA_SUCCESS;
        }
        std::vector<float> windowz(kNfft, 0.0f);
        for (int i = 0; i < cfg->win; ++i) windowz[off + i] = hann[i];
        p->edge_zero = !fa::sw_on(fa::Sw::MEL_NO_EZ);
        for (int i = 0; i < 32; ++i) if (windowz[i] != 0.0f || windowz[kNfft - 32 + i] != 0.0f) p->edge_zero = false;
        std::vector<float2> tw256(256), tw512(129);
        for (int k = 0; k < 256; ++k) { const double a = -2.0 * M_PI * k / 256.0; tw256[k] = make_float2((float)cos(a), (float)sin(a)); }
        for (int k = 0; k < 129; ++k) { const double a = -2.0 * M_PI * k / 512.0; tw512[k] = make_float2((float)cos(a), (float)sin(a)); }
        std::vector<int32_t> tab(cfg->n_mels);
        std::vector<float> weights;
        bool fast = cfg->n_mels <= kFastGroups * kGroup;
        std::vector<int> lo_of(cfg->n_mels, 0), cnt_of(cfg->n_mels, 0);
        for (int m = 0; m < cfg->n_mels; ++m) {
            int lo = -1, hi = -1;
            for (int k = 0; k < bins; ++k)
                if (fb[static_cast<size_t>(m) * bins + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
            const int cnt = lo < 0 ? 0 : hi - lo + 1;
            if (lo < 0) lo = 0;
            lo_of[m] = lo; cnt_of[m] = cnt;
            if (cnt > fast_slots(m / kGroup < kFastGroups ? m / kGroup : kFastGroups - 1)) fast = false;
        }
        if (fast) {  // [slot][lane] zero-padded weights; mel_tab keeps (lo, cnt) and a dummy start
            weights.assign(static_cast<size_t>(kFastSlots) * kGroup, 0.0f);
            for (int m = 0; m < cfg->n_mels; ++m) {
                const int i = m / kGroup, l = m % kGroup;
                for (int j = 0; j < cnt_of[m]; ++j)
                    weights[static_cast<size_t>(fast_slot_base(i) + j) * kGroup + l] = fb[static_cast<size_t>(m) * bins + lo_of[m] + j];
                tab[m] = lo_of[m] | (cnt_of[m] << 10);
            }
        } else {
            for (int m = 0; m < cfg->n_mels; ++m) {
                const int st = static_cast<int>(weights.size());
                for (int j = 0; j < cnt_of[m]; ++j) weights.push_back(fb[static_cast<size_t>(m) * bins + lo_of[m] + j]);
                tab[m] = lo_of[m] | (cnt_of[m] << 10) | (st << 20);
            }
        }
        p->fast = fast;
        if (weights.empty()) weights.push_back(0.0f);

        // device blob
        auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
        size_t o_off = 0, o_fr = align(o_off + sizeof(int64_t) * (batch + 1)), o_wz = align(o_fr + sizeof(int32_t) * batch),
               o_t256 = align(o_wz + sizeof(float) * kNfft), o_t512 = align(o_t256 + sizeof(float2) * 256),
               o_tab = align(o_t512 + sizeof(float2) * 129), o_w = align(o_tab + sizeof(int32_t) * cfg->n_mels),
               o_q = align(o_w + sizeof(float) * weights.size()), total = align(o_q + 8);
        std::vector<char> blob(total, 0);
        memcpy(blob.data() + o_off, offsets, sizeof(int64_t) * (batch + 1));
        memcpy(blob.data() + o_fr, frames.data(), sizeof(int32_t) * batch);
        memcpy(blob.data() + o_wz, windowz.data(), sizeof(float) * kNfft);
        memcpy(blob.data() + o_t256, tw256.data(), sizeof(float2) * 256);
        memcpy(blob.data() + o_t512, tw512.data(), sizeof(float2) * 129);
        memcpy(blob.data() + o_tab, tab.data(), sizeof(int32_t) * cfg->n_mels);
        memcpy(blob.data() + o_w, weights.data(), sizeof(float) * weights.size());
        hipError_t e = hipMalloc(&p->dev, total);
        if (e != hipSuccess) { delete p; return fa::hip_status(ctx, e, "mel plan hipMalloc"); }
        e = hipMemcpy(p->dev, blob.data(), total, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(p->dev); delete p; return fa::hip_status(ctx, e, "mel plan upload"); }
        char *d = static_cast<char *>(p->dev);

        MelArgs &a = p->args;
        a.offsets = reinterpret_cast<const int64_t *>(d + o_off);
        a.frames = reinterpret_cast<const int32_t *>(d + o_fr);
        a.windowz = reinterpret_cast<const float *>(d + o_wz);
        a.tw256 = reinterpret_cast<const float2 *>(d + o_t256);
        a.tw512 = reinterpret_cast<const float2 *>(d + o_t512);
        a.mel_tab = reinterpret_cast<const int32_t *>(d + o_tab);
        a.mel_w = reinterpret_cast<const float *>(d + o_w);
        a.queue = reinterpret_cast<unsigned long long *>(d + o_q);   // zero in the blob
        a.utt_stride = p->utt_stride;
        a.tiles_per_utt = (frame_stride + kTileFrames - 1) / kTileFrames;
        a.total_tiles = static_cast<int64_t>(a.tiles_per_utt) * batch;
        if (a.total_tiles > INT32_MAX) { (void)hipFree(p->dev); delete p; return fa::set_error(ctx, FA_INDEX_OVERFLOW, "mel: too many tiles in one plan"); }
        a.frame_stride = frame_stride;
        a.n_mels = cfg->n_mels;
        a.n_weights = static_cast<int32_t>(weights.size());
        a.hop = cfg->hop;
        a.pad = cfg->padding_mode == FA_MEL_PAD_CENTER ? cfg->n_fft / 2 : 0;
        a.stage_count = (kTileFrames - 1) * cfg->hop + kNfft;
        a.stage_alloc = (a.stage_count + 3) & ~3;
        {
            const int mm = cfg->n_mels * kMelPad, fm = kTileFrames * (cfg->n_mels + kFramePad);
            a.out_alloc = ((mm > fm ? mm : fm) + 3) & ~3;
        }
        a.preemph = cfg->padding_mode == FA_MEL_PAD_LEGACY ? 0.0f : cfg->preemph;  // compute() has no pre-emphasis (:146-153)
        a.log_floor = cfg->log_floor;
        a.floor_clamped = cfg->floor_mode == FA_MEL_FLOOR_CLAMPED;
        a.prio_lo = 0; a.prio_hi = 3; a.prio_pw = 1; a.prio_rd = 2;   // measured best of the sweep in DESIGN.md §3.1
        if (const char *pe = fa::sw(fa::Sw::MEL_PRIO)) {   // diagnostics
            int v4[4] = {0, 3, 1, 2};
            const int got = sscanf(pe, "%d,%d,%d,%d", &v4[0], &v4[1], &v4[2], &v4[3]);
            if (got == 2) { v4[2] = v4[0]; v4[3] = v4[1]; }
            if (got >= 2) { a.prio_lo = v4[0] & 3; a.prio_hi = v4[1] & 3; a.prio_pw = v4[2] & 3; a.prio_rd = v4[3] & 3; }
        }
        p->pk = fast && cfg->hop == kPkHop && !fa::sw_on(fa::Sw::MEL_SCALAR);   // FA_MEL_SCALAR: diagnostics, one frame per lane
        if (p->pk && cfg->n_mels == kFastGroups * kGroup) {   // FA_MEL_V4=0: the v3 kernel (diagnostics); FA_MEL_V4=4: four workgroups per CU
            const char *ve = fa::sw(fa::Sw::MEL_V4);
            p->v4_wps = ve ? atoi(ve) : 3;
            if (p->v4_wps != 3) p->v4_wps = 0;
            if (const char *de = fa::sw(fa::Sw::MEL_V4_DEEP)) p->v4_deep = atoi(de) != 0;
        }
        p->lds_bytes = sizeof(float) * (a.stage_alloc + kRegions * (p->pk ? kRegionFloatsPk : kRegionFloats) + a.out_alloc) + sizeof(int32_t) * kMaxMels +
                       sizeof(float) * (static_cast<size_t>(a.n_weights) + 24 + 4 + (p->pk ? fa::melpk::kWindowTableFloats : 0));   // the paired weight reads of the packed kernel touch one slot row past the table
        if (p->v4_wps) {
            p->lds_bytes = kV4LdsBytes;
            if (const char *pe = fa::sw(fa::Sw::MEL_V4_LDS_PAD)) p->lds_bytes += static_cast<size_t>(atoi(pe));   // diagnostics: fewer resident workgroups per CU
        }
        if (p->lds_bytes > 160 * 1024) { (void)hipFree(p->dev); delete p; return fa::set_error(ctx, FA_INVALID_ARGUMENT, "mel: hop too large for LDS staging"); }
        if (p->lds_bytes > 64 * 1024) {
            const int lb = static_cast<int>(p->lds_bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<0, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<1, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<0, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<1, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<0, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<0, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<1, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<1, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
#define FA_V4_ATTR(L, E) do { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel_v4<L, E, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lb); \
                              (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel_v4<L, E, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb); \
                              (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel_v4<L, E, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lb); \
                              (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel_v4<L, E, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb); } while (0)
            FA_V4_ATTR(0, true); FA_V4_ATTR(0, false); FA_V4_ATTR(1, true); FA_V4_ATTR(1, false);
#undef FA_V4_ATTR
        }
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, ctx->device);
        const int cus = e == hipSuccess ? prop.multiProcessorCount : 256;
        const char *rounds_env = fa::sw(fa::Sw::MEL_ROUNDS);  // diagnostics
        // Equal-length batches: one persistent round (the per-workgroup prologue — tables, lane constants — is paid once:
        // 0.663 vs 0.685 ms with four rounds on the bench workload).  Ragged batches keep four rounds, so that the hardware
        // scheduler evens out ranges that hold many empty tiles of short utterances.
        bool uniform = true;
        for (int b = 1; b < batch; ++b) if (frames[b] != frames[0]) { uniform = false; break; }
        const int rounds = rounds_env && atoi(rounds_env) > 0 ? atoi(rounds_env) : (uniform ? 1 : 4);
        const int64_t want = static_cast<int64_t>(cus) * (p->v4_wps ? p->v4_wps : 2) * rounds;  // resident workgroups per CU, `rounds` rounds of them
        p->grid = static_cast<int>(a.total_tiles < want ? a.total_tiles : want);
        if (p->grid < 1) p->grid = 1;
        *out = p;
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "mel plan: host allocation failed");
    } catch (...) {
        return fa::set_error(ctx, FA_UNKNOWN_ERROR, "mel plan: unexpected failure");
    }
}

void fa_mel_plan_destroy(fa_mel_plan *p) {
    if (!p) return;
    if (p->dev) { (void)hipSetDevice(p->ctx->device); (void)hipStreamSynchronize(p->ctx->stream); (void)hipFree(p->dev); }
    delete p;
}

int64_t fa_mel_plan_utt_stride(const fa_mel_plan *p) { return p ? p->utt_stride : 0; }
int32_t fa_mel_plan_frame_stride(const fa_mel_plan *p) { return p ? p->frame_stride : 0; }
int64_t fa_mel_plan_total_frames(const fa_mel_plan *p) { return p ? p->total_frames : 0; }

fa_status fa_mel_execute_dev(fa_mel_plan *p, const float *d_pcm, const float *d_last, float *d_mel, int32_t *d_lengths) {
    if (!p || !d_mel || (!d_pcm && p->total_samples > 0)) return FA_INVALID_ARGUMENT;
    fa_ctx *ctx = p->ctx;
    fa::DeviceGuard guard(ctx->device);
    if (p->generic) {
        fa::melgen::GenArgs g = p->gargs;
        g.pcm = d_pcm; g.last = d_last; g.out = d_mel; g.lengths = d_lengths;
        hipLaunchKernelGGL(fa::melgen::mel_generic_kernel, dim3(p->grid), dim3(fa::melgen::kThreads), p->lds_bytes, ctx->stream, g);
        FA_HIP_TRY(ctx, hipGetLastError());
        return FA_SUCCESS;
    }
    MelArgs a = p->args;
    a.pcm = d_pcm; a.last = d_last; a.out = d_mel; a.lengths = d_lengths;
    static unsigned long long *s_prof = nullptr;
    static int s_prof_calls = 0;
    static unsigned long long s_last_span[4] = {0, 0, 0, 0};
    static std::mutex s_prof_mutex;   // the diagnostics state is process-wide; entries of different contexts may run concurrently
    std::unique_lock<std::mutex> prof_lock(s_prof_mutex, std::defer_lock);
    if (fa::sw(fa::Sw::MEL_PROF)) {  // diagnostics only: per-phase cycles of one workgroup, printed every 10 launches
        prof_lock.lock();
        if (!s_prof) { (void)hipMalloc(&s_prof, 128 + 4 * 8192); (void)hipMemset(s_prof, 0, 128 + 4 * 8192); }
        a.prof = s_prof;
        {   // slots 2..5: min / max of the workgroup start and end times of the launch about to be made
            const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
            unsigned long long keep[4];
            (void)hipMemcpy(keep, s_prof + 2, 32, hipMemcpyDeviceToHost);
            (void)hipMemcpy(s_prof + 2, init, 32, hipMemcpyHostToDevice);
            if (s_prof_calls % 10 == 9) memcpy(s_last_span, keep, 32);
        }
        if (++s_prof_calls % 10 == 0) {
            unsigned long long h[16];
            (void)hipMemcpy(h, s_prof, 128, hipMemcpyDeviceToHost);
            const double n = h[7] ? static_cast<double>(h[7]) : 1.0;
            if (p->v4_wps) {
                fprintf(stderr, "mel v4 profile (cycles per tile, wave 0 of one workgroup, %llu tiles): stage %.0f | B1 %.0f | reads+pass1 %.0f | B2 %.0f | pass2..log %.0f | B3 %.0f | loads+stores %.0f | B4 %.0f\n",
                        h[7], h[8] / n, h[9] / n, h[10] / n, h[11] / n, h[12] / n, h[13] / n, h[14] / n, h[15] / n);
                fprintf(stderr, "mel v4 profile: workgroup starts spread over %.1f us, ends over %.1f us, first start -> last end %.1f us (last launch)\n",
                        (s_last_span[1] - s_last_span[0]) / 100.0, (s_last_span[3] - s_last_span[2]) / 100.0, (s_last_span[3] - s_last_span[0]) / 100.0);
                {
                    std::vector<unsigned> dur(p->grid);
                    (void)hipMemcpy(dur.data(), s_prof + 16, sizeof(unsigned) * dur.size(), hipMemcpyDeviceToHost);
                    std::vector<unsigned> srt(dur);
                    std::sort(srt.begin(), srt.end());
                    fprintf(stderr, "mel v4 profile: workgroup durations (us): min %.1f | p10 %.1f | median %.1f | p90 %.1f | max %.1f ; slowest blocks:", srt.front() / 100.0,
                            srt[srt.size() / 10] / 100.0, srt[srt.size() / 2] / 100.0, srt[srt.size() * 9 / 10] / 100.0, srt.back() / 100.0);
                    int shown = 0;
                    for (size_t b = 0; b < dur.size() && shown < 12; ++b) if (dur[b] >= srt[srt.size() - 12]) { fprintf(stderr, " %zu", b); ++shown; }
                    double byx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (size_t b = 0; b < dur.size(); ++b) byx[b & 7] += dur[b];
                    fprintf(stderr, "\n   mean by 64-block bucket (us):");
                    for (size_t b0 = 0; b0 < dur.size(); b0 += 64) { double m = 0; size_t c = 0; for (size_t b = b0; b < b0 + 64 && b < dur.size(); ++b, ++c) m += dur[b]; fprintf(stderr, " %.0f", m / c / 100.0); }
                    fprintf(stderr, "\n   mean by blockIdx %% 8 (us):");
                    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", byx[x] / (dur.size() / 8.0) / 100.0);
                    fprintf(stderr, "\n");
                }
                if (h[1]) fprintf(stderr, "mel v4 profile: shader clock during the kernel %.0f MHz (clock64 / wall_clock64 at 100 MHz)\n", 100.0 * static_cast<double>(h[0]) / static_cast<double>(h[1]));
            } else {
            fprintf(stderr, "mel profile (cycles per tile, wave 0 of one workgroup, %llu tiles): stage-write %.0f | barrier1 %.0f | prefetch issue %.0f | passes %.0f | barrier2 %.0f | store %.0f\n",
                    h[7], h[0] / n, h[1] / n, (h[2] + h[6]) / n, (h[3] + h[8] + h[9] + h[10] + h[11]) / n, h[4] / n, h[5] / n);
            if (h[8]) fprintf(stderr, "mel profile, packed pass: sample reads %.0f | fft256 %.0f | partner + power %.0f | filterbank %.0f | log + stage %.0f\n",
                              h[8] / n, h[9] / n, h[10] / n, h[11] / n, h[3] / n);
            }
        }
    } else a.prof = nullptr;
    const bool mm = p->cfg.layout == FA_MEL_LAYOUT_MEL_MAJOR;
    const bool pk = p->pk;
    const dim3 grid(p->grid), block(kThreads);
    if (p->v4_wps) {
        // every workgroup draws one index per tile it processes plus the one that tells it to stop: a launch advances the
        // counter by exactly total_tiles + grid
        a.queue_base = p->launches++ * (static_cast<unsigned long long>(a.total_tiles) + static_cast<unsigned long long>(p->v4_deep ? 2 : 1) * static_cast<unsigned long long>(p->grid));
        const bool ez = p->edge_zero, w4 = p->v4_deep;   // w4: the deep-queue variant
#define FA_V4(L, E, D) do { if (p->cfg.floor_mode == FA_MEL_FLOOR_CLAMPED) hipLaunchKernelGGL((mel_kernel_v4<L, E, D, true>), grid, block, p->lds_bytes, ctx->stream, a); \
                             else hipLaunchKernelGGL((mel_kernel_v4<L, E, D, false>), grid, block, p->lds_bytes, ctx->stream, a); } while (0)
        if (mm) { if (ez) { if (w4) FA_V4(FA_MEL_LAYOUT_MEL_MAJOR, true, true); else FA_V4(FA_MEL_LAYOUT_MEL_MAJOR, true, false); }
                  else { if (w4) FA_V4(FA_MEL_LAYOUT_MEL_MAJOR, false, true); else FA_V4(FA_MEL_LAYOUT_MEL_MAJOR, false, false); } }
        else { if (ez) { if (w4) FA_V4(FA_MEL_LAYOUT_FRAME_MAJOR, true, true); else FA_V4(FA_MEL_LAYOUT_FRAME_MAJOR, true, false); }
               else { if (w4) FA_V4(FA_MEL_LAYOUT_FRAME_MAJOR, false, true); else FA_V4(FA_MEL_LAYOUT_FRAME_MAJOR, false, false); } }
#undef FA_V4
        FA_HIP_TRY(ctx, hipGetLastError());
        return FA_SUCCESS;
    }
    if (mm && pk && p->edge_zero) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_MEL_MAJOR, true, 2>), grid, block, p->lds_bytes, ctx->stream, a);
    else if (pk && p->edge_zero) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_FRAME_MAJOR, true, 2>), grid, block, p->lds_bytes, ctx->stream, a);
    else if (mm && pk) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_MEL_MAJOR, true, 1>), grid, block, p->lds_bytes, ctx->stream, a);
    else if (pk) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_FRAME_MAJOR, true, 1>), grid, block, p->lds_bytes, ctx->stream, a);
    else if (mm && p->fast) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_MEL_MAJOR, true, 0>), grid, block, p->lds_bytes, ctx->stream, a);
    else if (mm) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_MEL_MAJOR, false, 0>), grid, block, p->lds_bytes, ctx->stream, a);
    else if (p->fast) hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_FRAME_MAJOR, true, 0>), grid, block, p->lds_bytes, ctx->stream, a);
    else hipLaunchKernelGGL((mel_kernel<FA_MEL_LAYOUT_FRAME_MAJOR, false, 0>), grid, block, p->lds_bytes, ctx->stream, a);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

fa_status fa_mel_normalize_per_feature_dev(fa_ctx *ctx, float *d_mel, int32_t batch, int32_t n_mels, int32_t frame_stride,
                                           int32_t frames, const int32_t *d_valid_frames) {
    if (!ctx || !d_mel || !d_valid_frames) return FA_INVALID_ARGUMENT;
    if (batch < 0 || n_mels < 1 || frames < 0 || frame_stride < frames) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "mel normalise: bad shape");
    if (batch == 0 || frames == 0) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    const int64_t rows = static_cast<int64_t>(batch) * n_mels;
    hipLaunchKernelGGL(mel_norm_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, ctx->stream, d_mel, d_valid_frames, rows,
                       n_mels, frame_stride, frames);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

// Plans of small host-pointer calls, kept per context (see fa_mel_batch).  Key = everything fa_mel_plan_create looks at.
}  // extern "C"
namespace {
constexpr int32_t kMelCacheMaxBatch = 8;
constexpr size_t kMelCacheEntries = 8;
constexpr size_t kMelCfgKeyBytes = offsetof(fa_mel_config, tail_mode) + sizeof(int32_t);   // every field in front of the filterbank pointer, no padding
struct MelPlanCache {
    struct Entry {
        fa_mel_config cfg;
        int32_t batch, frame_stride;
        bool has_expected;
        std::vector<int64_t> offsets;
        std::vector<int32_t> expected;
        fa_mel_plan *plan;
    };
    std::vector<Entry> entries;
};
void mel_cache_free(void *p) {
    MelPlanCache *c = static_cast<MelPlanCache *>(p);
    if (!c) return;
    for (auto &en : c->entries) fa_mel_plan_destroy(en.plan);
    delete c;
}
fa_status mel_cached_plan(fa_ctx *ctx, const fa_mel_config *cfg, const int64_t *offsets, int32_t batch, const int32_t *expected_frames,
                          int32_t frame_stride, fa_mel_plan **out) {
    try {
        if (!ctx->mel_cache) { ctx->mel_cache = new MelPlanCache; ctx->mel_cache_free = mel_cache_free; }
        MelPlanCache *c = static_cast<MelPlanCache *>(ctx->mel_cache);
        for (auto &en : c->entries) {
            if (en.batch != batch || en.frame_stride != frame_stride || en.has_expected != (expected_frames != nullptr)) continue;
            if (memcmp(&en.cfg, cfg, kMelCfgKeyBytes) != 0) continue;
            if (memcmp(en.offsets.data(), offsets, sizeof(int64_t) * (batch + 1)) != 0) continue;
            if (expected_frames && memcmp(en.expected.data(), expected_frames, sizeof(int32_t) * batch) != 0) continue;
            *out = en.plan;
            return FA_SUCCESS;
        }
        fa_mel_plan *p = nullptr;
        FA_TRY(fa_mel_plan_create(ctx, cfg, offsets, batch, expected_frames, frame_stride, &p));
        if (c->entries.size() >= kMelCacheEntries) { fa_mel_plan_destroy(c->entries.front().plan); c->entries.erase(c->entries.begin()); }
        MelPlanCache::Entry en;
        en.cfg = *cfg; en.batch = batch; en.frame_stride = frame_stride; en.has_expected = expected_frames != nullptr;
        en.offsets.assign(offsets, offsets + batch + 1);
        if (expected_frames) en.expected.assign(expected_frames, expected_frames + batch);
        en.plan = p;
        c->entries.push_back(std::move(en));
        *out = p;
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "mel plan cache: host allocation failed");
    }
}
}  // namespace
extern "C" {

// Host-pointer entry.  Pageable host buffers: copy in, kernel, copy out on the context's stream (each copy is a staged copy inside
// the runtime at ~55 GB/s; measured: slicing + a second host thread does not overlap the two directions, the staging serialises).
// PINNED host buffers (fa_host_alloc, or memory the caller registered with the HIP runtime): the batch is cut into slices of
// utterances (~64 MB of samples each), slice k + 1 is uploaded and launched on the context's stream while the log-mel of slice k
// is downloaded on a second stream — true DMA in both directions of the PCIe link at once.  One slice = one plan.
fa_status fa_mel_batch(fa_ctx *ctx, const fa_mel_config *cfg, const float *pcm, const int64_t *offsets, int32_t batch,
                       const float *last_samples, const int32_t *expected_frames, int32_t frame_stride, float *mel,
                       int32_t *mel_lengths) {
    if (!ctx || !mel || !offsets || batch < 1) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(ctx->device);
    fa_mel_plan *whole = nullptr;   // geometry of the whole batch (frame stride, utterance stride) + validation
    // Small calls (the reference's streaming callers: one chunk of a fixed length per call, StreamingEouAsrManager.swift:558) keep their
    // plan in the context: building the tables on the host, one hipMalloc / upload / hipFree for them and three more for the I/O buffers
    // were 110 of the 194 us such a call took (scripts/mel_latency_probe.py).
    const bool cached = batch <= kMelCacheMaxBatch && cfg && cfg->filterbank == nullptr;
    if (cached) FA_TRY(mel_cached_plan(ctx, cfg, offsets, batch, expected_frames, frame_stride, &whole));
    else FA_TRY(fa_mel_plan_create(ctx, cfg, offsets, batch, expected_frames, frame_stride, &whole));
    struct PlanOwner { fa_mel_plan *&p; bool own; ~PlanOwner() { if (own && p) fa_mel_plan_destroy(p); } } owner{whole, !cached};
    const int64_t ns = offsets[batch];
    if (ns > 0 && !pcm) return FA_INVALID_ARGUMENT;
    const int32_t fstride = whole->frame_stride;
    const int64_t ustride = whole->utt_stride;
    auto pinned = [](const void *p) {
        hipPointerAttribute_t at{};
        if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    int64_t slice_bytes = pinned(pcm) && pinned(mel) ? (64ll << 20) : (1ll << 62);
    if (const char *se = fa::sw(fa::Sw::MEL_SLICE_MB)) { const long v = atol(se); slice_bytes = v > 0 ? v * (1ll << 20) : (1ll << 62); }   // diagnostics / tests; 0 = one slice
    // slices: consecutive utterances up to slice_bytes of samples or of output, whichever is reached first
    std::vector<int32_t> first{0};
    for (int32_t b = 0; b < batch; ++b) {
        const int32_t f = first.back();
        const int64_t in_bytes = 4 * (offsets[b + 1] - offsets[f]), out_bytes = 4 * ustride * (b + 1 - f);
        if (b + 1 < batch && (in_bytes >= slice_bytes || out_bytes >= slice_bytes)) first.push_back(b + 1);
    }
    first.push_back(batch);
    const int n_slices = static_cast<int>(first.size()) - 1;
    const size_t out_floats = static_cast<size_t>(ustride) * batch;
    hipError_t e = hipSuccess;
    if (n_slices <= 1) {   // one slice: samples, output and lengths in the context's scratch buffer (grow-only, kept between calls)
        auto al = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
        const size_t o_pcm = 0, o_out = al(sizeof(float) * static_cast<size_t>(ns)), o_len = o_out + al(sizeof(float) * out_floats),
                     o_last = o_len + al(sizeof(int32_t) * batch), total = o_last + al(sizeof(float) * batch);
        if (fa::ensure_scratch(ctx, total) != FA_SUCCESS) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "fa_mel_batch: device allocation failed"); }
        char *base = static_cast<char *>(ctx->scratch);
        float *d_pcm = reinterpret_cast<float *>(base + o_pcm), *d_out = reinterpret_cast<float *>(base + o_out), *d_last = reinterpret_cast<float *>(base + o_last);
        int32_t *d_len = reinterpret_cast<int32_t *>(base + o_len);
        fa_status st = FA_SUCCESS;
        do {
            if (ns > 0 && (e = hipMemcpyAsync(d_pcm, pcm, sizeof(float) * ns, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            if (last_samples && (e = hipMemcpyAsync(d_last, last_samples, sizeof(float) * batch, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            st = fa_mel_execute_dev(whole, d_pcm, last_samples ? d_last : nullptr, d_out, d_len);
            if (st != FA_SUCCESS) break;
            if ((e = hipMemcpyAsync(mel, d_out, sizeof(float) * out_floats, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
            if (mel_lengths && (e = hipMemcpyAsync(mel_lengths, d_len, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
            e = hipStreamSynchronize(ctx->stream);
        } while (0);
        if (st != FA_SUCCESS) return st;
        return fa::hip_status(ctx, e, "fa_mel_batch");
    }
    fa::DevBuf d_pcm, d_last, d_out, d_len;
    e = d_pcm.alloc(sizeof(float) * static_cast<size_t>(ns));
    if (e == hipSuccess) e = d_out.alloc(sizeof(float) * out_floats);
    if (e == hipSuccess) e = d_len.alloc(sizeof(int32_t) * batch);
    if (e == hipSuccess && last_samples) e = d_last.alloc(sizeof(float) * batch);
    if (e != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "fa_mel_batch: device allocation failed"); }
    if (owner.own) { fa_mel_plan_destroy(whole); whole = nullptr; }   // only its geometry was needed: every slice gets its own plan
    try {
        hipStream_t down = nullptr;
        FA_HIP_TRY(ctx, hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
        std::vector<hipEvent_t> done(n_slices, nullptr);
        std::vector<fa_mel_plan *> plans(n_slices, nullptr);
        fa_status st = FA_SUCCESS;
        if (last_samples && (e = hipMemcpyAsync(d_last.p, last_samples, sizeof(float) * batch, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) st = fa::hip_status(ctx, e, "fa_mel_batch upload");
        for (int k = 0; k < n_slices && st == FA_SUCCESS; ++k) {
            const int32_t f = first[k], cnt = first[k + 1] - f;
            std::vector<int64_t> offs(cnt + 1);
            for (int32_t i = 0; i <= cnt; ++i) offs[i] = offsets[f + i] - offsets[f];
            st = fa_mel_plan_create(ctx, cfg, offs.data(), cnt, expected_frames ? expected_frames + f : nullptr, fstride, &plans[k]);
            if (st != FA_SUCCESS) break;
            hipError_t ue = hipSuccess;
            if (offs[cnt] > 0) ue = hipMemcpyAsync(d_pcm.as<float>() + offsets[f], pcm + offsets[f], sizeof(float) * offs[cnt], hipMemcpyHostToDevice, ctx->stream);
            if (ue == hipSuccess) st = fa_mel_execute_dev(plans[k], d_pcm.as<float>() + offsets[f], last_samples ? d_last.as<float>() + f : nullptr,
                                                          d_out.as<float>() + static_cast<size_t>(ustride) * f, d_len.as<int32_t>() + f);
            if (ue == hipSuccess && st == FA_SUCCESS) ue = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
            if (ue == hipSuccess && st == FA_SUCCESS) ue = hipEventRecord(done[k], ctx->stream);
            if (ue == hipSuccess && st == FA_SUCCESS) ue = hipStreamWaitEvent(down, done[k], 0);
            if (ue == hipSuccess && st == FA_SUCCESS)
                ue = hipMemcpyAsync(mel + static_cast<size_t>(ustride) * f, d_out.as<float>() + static_cast<size_t>(ustride) * f,
                                    sizeof(float) * static_cast<size_t>(ustride) * cnt, hipMemcpyDeviceToHost, down);
            if (ue != hipSuccess) st = fa::hip_status(ctx, ue, "fa_mel_batch slice");
        }
        if (st == FA_SUCCESS && mel_lengths && (e = hipMemcpyAsync(mel_lengths, d_len.p, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
            st = fa::hip_status(ctx, e, "fa_mel_batch lengths");
        const hipError_t s1 = hipStreamSynchronize(ctx->stream), s2 = hipStreamSynchronize(down);
        for (auto *pl : plans) if (pl) fa_mel_plan_destroy(pl);
        for (auto ev : done) if (ev) (void)hipEventDestroy(ev);
        (void)hipStreamDestroy(down);
        if (st != FA_SUCCESS) return st;
        return fa::hip_status(ctx, s1 != hipSuccess ? s1 : s2, "fa_mel_batch");
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "fa_mel_batch: host allocation failed");
    } catch (...) {
        return fa::set_error(ctx, FA_UNKNOWN_ERROR, "fa_mel_batch: unexpected failure");
    }
}

}  // extern "C"
