// resample.hip — sample-rate conversion to the 16 kHz mono fp32 the featurizer consumes (gfx950).
//
// (1) fa_resample_linear replaces AudioConverter.linearResample
//     (reference: Sources/FluidAudio/Shared/AudioConverter.swift:388-442): mix N planar channels down to mono with
//     weight 1/N (:399-408), then linear interpolation at sourceIndex = i * (inRate / outRate) in double precision
//     (:419-434).  This is the only resampling arithmetic that exists in the reference tree; it is pinned by
//     AudioConverterTests.swift:546-761 and reproduced bit-for-bit (fp32 mix and blend with one rounding per operation).
// (2) fa_resample_poly is an EXTENSION with its own specification (PARITY UNPINNED): the reference's default path hands
//     the job to Apple's closed-source AVAudioConverter (AudioConverter.swift:299-370), whose arithmetic cannot be
//     restated.  The kernel is a rational-ratio polyphase FIR (Kaiser beta = 5 windowed sinc, half length
//     10 * max(up, down), unit DC gain, the output alignment of scipy.signal.resample_poly) so that an independent
//     second opinion exists on the CPU.  One thread per output sample walks the ~2*10*max(up,down)/up taps of its
//     phase; consecutive threads read consecutive input samples and taps `up` apart.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fa_common.h"
#include "resample_geom.h"

namespace {

using fa::PolyRowsGeom;
using fa::kRowsThreads;
using fa::kRowsWaves;
using fa::kRowsOffLane;
constexpr int kThreads = 256;

__global__ void mixdown_kernel(const float *__restrict__ planar, float *__restrict__ mono, int channels, int64_t frames) {
    const int64_t f = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    float sum = 0.0f;
    for (int c = 0; c < channels; ++c) sum = __fadd_rn(sum, planar[static_cast<int64_t>(c) * frames + f]);  // :403-407
    mono[f] = __fmul_rn(sum, 1.0f / static_cast<float>(channels));
}

__global__ void linear_kernel(const float *__restrict__ mono, float *__restrict__ out, int64_t frames, int64_t out_frames, double ratio) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= out_frames) return;
    const double src = static_cast<double>(i) * ratio;       // :424
    const int64_t idx = static_cast<int64_t>(src);            // Int(sourceIndex): truncation
    const float frac = static_cast<float>(src - static_cast<double>(idx));
    float v = 0.0f;
    if (idx < frames - 1) v = __fadd_rn(__fmul_rn(mono[idx], __fsub_rn(1.0f, frac)), __fmul_rn(mono[idx + 1], frac));  // :428-430
    else if (idx < frames) v = mono[idx];                     // :431-432
    out[i] = v;
}

__global__ void poly_kernel(const float *__restrict__ x, const float *__restrict__ h, float *__restrict__ y, int64_t n_in, int64_t n_out,
                            int64_t h_len, int up, int down, int64_t pre_remove, int64_t m_lo, int64_t m_hi) {
    const int64_t m = m_lo + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (m >= m_hi || m >= n_out) return;
    const int64_t p = (m + pre_remove) * down;  // position in the zero-stuffed stream
    int64_t k_hi = p / up;                      // largest k with p - k*up >= 0
    int64_t k_lo = (p - (h_len - 1) + up - 1) / up;  // smallest k with p - k*up <= h_len - 1
    if (p - (h_len - 1) < 0) k_lo = 0;
    if (k_hi > n_in - 1) k_hi = n_in - 1;
    float acc = 0.0f;
    for (int64_t k = k_lo; k <= k_hi; ++k) acc = fmaf(h[p - k * up], x[k], acc);
    y[m] = acc;
}

// LDS-staged polyphase FIR: persistent workgroups keep ALL taps in LDS (loaded once per workgroup), walk tiles of kPolyTile
// outputs, and stage the input span a tile needs (tile * down / up + h_len / up samples) in LDS with coalesced loads: every
// input sample is read from HBM once per tile it touches, every output written once; the h_len / up multiply-adds of an output
// read both operands from LDS.  Same summation order (ascending input index) as poly_kernel: bit-identical results.
// Algorithmic HBM bytes = 4 (n_in + n_out).
constexpr int kPolyTile = 2048;

__global__ __launch_bounds__(kThreads) void poly_lds_kernel(const float *__restrict__ x, const float *__restrict__ h, float *__restrict__ y, int64_t n_in,
                                                            int64_t n_out, int h_len, int up, int down, int64_t pre_remove, int span_alloc) {
    extern __shared__ float sm[];
    float *taps = sm, *xs = sm + ((h_len + 3) & ~3);
    for (int i = threadIdx.x; i < h_len; i += kThreads) taps[i] = h[i];
    const int64_t tiles = (n_out + kPolyTile - 1) / kPolyTile;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t m0 = tile * kPolyTile;
        const int64_t m1 = m0 + kPolyTile < n_out ? m0 + kPolyTile : n_out;
        const int64_t p_first = (m0 + pre_remove) * down, p_last = (m1 - 1 + pre_remove) * down;
        int64_t k0 = p_first - (h_len - 1) < 0 ? 0 : (p_first - (h_len - 1) + up - 1) / up;
        int64_t k1 = p_last / up;
        if (k1 > n_in - 1) k1 = n_in - 1;
        const int span = k1 >= k0 ? static_cast<int>(k1 - k0 + 1) : 0;
        __syncthreads();   // the previous tile's reads of xs are complete (and the taps are in place)
        for (int i = threadIdx.x; i < span && i < span_alloc; i += kThreads) xs[i] = x[k0 + i];
        __syncthreads();
        for (int64_t m = m0 + threadIdx.x; m < m1; m += kThreads) {
            const int64_t p = (m + pre_remove) * down;
            int64_t k_hi = p / up;
            int64_t k_lo = p - (h_len - 1) < 0 ? 0 : (p - (h_len - 1) + up - 1) / up;
            if (k_hi > n_in - 1) k_hi = n_in - 1;
            int ti = static_cast<int>(p - k_lo * up);       // tap of the first term, then -up per input sample
            const float *xp = xs + (k_lo - k0);
            const int cnt = static_cast<int>(k_hi - k_lo + 1);
            float acc = 0.0f;
            for (int j = 0; j < cnt; ++j, ti -= up) acc = fmaf(taps[ti], xp[j], acc);
            y[m] = acc;
        }
    }
}

// Integer decimation (up == 1, down 2 .. 5: 32 / 48 / 64 / 80 kHz -> 16 kHz), register-tiled: every output has the SAME taps, so they are read
// once per wavefront through the scalar cache (constant address space -> SGPRs, folded into the fused multiply-adds as scalar
// operands), and R consecutive outputs of a thread share their inputs: NT + (R - 1) DOWN samples fetched with 16-byte loads straight
// into registers — no LDS at all.  The LDS-staged kernel above spends two LDS operand reads per multiply-add (11.8 % of the HBM
// roofline for 48 -> 16 kHz, LDS-issue bound); here a multiply-add costs one v_fmac with a scalar operand.  Per output the terms are
// added in ascending input order with fmaf, INCLUDING the leading zero taps: bit-identical to poly_kernel.
// Geometry for up == 1 (fa_resample_poly_taps): half = 10 DOWN, NT = 21 DOWN + 1 taps of which the first DOWN are zeros,
// pre_remove = 11: output m reads inputs (m - 10) DOWN ... (m + 11) DOWN with tap (m + 11) DOWN - k.  The launch covers outputs
// [m_begin, m_begin + count R) whose inputs all exist; m_begin = 10 (mod 4) and R DOWN = 0 (mod 4) make every thread's first input a
// multiple of 4 samples: aligned 16-byte loads with a compile-time lane layout.  The edges go to poly_kernel.
// A tap as the DPP operand of the multiply-add (round 5): v_fmac_f32_dpp ... row_newbcast:K multiplies with lane K of the lane's own 16-lane row of
// `taps16`, so a register holds 16 taps (replicated in its four rows) and a multiply-add with a wave-uniform tap is ONE vector instruction without
// a scalar register per tap (down = 6 has 127 taps: more than the scalar file holds as asm operands) and without a v_readlane per tap.
template <int K>
__device__ __forceinline__ void fmac_bcast(float &acc, const float taps16, const float x) {   // acc += taps16[lane K of this lane's row] * x
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(taps16), "v"(x), "n"(K));
}
// a wave-uniform tap as the SCALAR operand of the multiply-add (written out: left to the compiler, the SLP vectoriser pairs outputs into v_pk_fma_f32 + register moves)
__device__ __forceinline__ void fmac_scalar(float &acc, const float tap, const float x) { asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(tap), "v"(x)); }
// compile-time loop (the DPP lane is an immediate of the instruction: the index must be a constant expression, not a value the unroller may or may not
// fold — the 169 x 8 body of down = 6 is beyond what it unrolls)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

constexpr int kDecimR = 8;
template <int DOWN>
__global__ __launch_bounds__(kThreads) void poly_decim_kernel(const float *__restrict__ x, const float *__restrict__ h, float *__restrict__ y, int64_t m_begin,
                                                              int64_t groups) {
    constexpr int R = kDecimR, NT = 21 * DOWN + 1, NIN = NT + (R - 1) * DOWN, NV = (NIN + 3) / 4;
    static_assert((R * DOWN) % 4 == 0, "aligned 16-byte loads need R * DOWN = 0 (mod 4)");
    const int64_t g_own = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    // every lane of a wavefront stays active to the end: the DPP form of the multiply-add reads its tap from ANOTHER lane of the row, and a lane
    // that has left reads as zero — the threads behind the last group recompute that group and skip the store
    const bool mine = g_own < groups;
    const int64_t g = mine ? g_own : groups - 1;
    const int64_t m0 = m_begin + g * R;
    const float4 *src = reinterpret_cast<const float4 *>(x + (m0 - 10) * DOWN);
    float xin[4 * NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { const float4 q = src[v]; xin[4 * v] = q.x; xin[4 * v + 1] = q.y; xin[4 * v + 2] = q.z; xin[4 * v + 3] = q.w; }
    typedef const float __attribute__((address_space(4))) *c_f32;
    const c_f32 taps = (c_f32)h;
    constexpr int NQ = (NT + 15) / 16;
    float tq[NQ];                                   // DOWN >= 4: the taps in vector registers, 16 per register, for the DPP form of the multiply-add
    if constexpr (DOWN >= 4) {
        const int l16 = threadIdx.x & 15;
#pragma unroll
        for (int qv = 0; qv < NQ; ++qv) { const int ti = 16 * qv + l16; tq[qv] = h[ti < NT ? ti : NT - 1]; }
    }
    float acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = 0.0f;
    // One v_fmac_f32 per multiply-add, the tap as its scalar operand — written as inline assembly since round 4: left to the compiler, the SLP
    // vectoriser paired the outputs into v_pk_fma_f32, whose operands must be consecutive register PAIRS; the inputs of two outputs lie DOWN
    // registers apart, so every packed instruction came with ~1.3 register moves (DOWN = 3: 256 v_pk_fma + 361 moves = 774 instructions for 8
    // outputs; now 512 v_fmac + the loads and stores).  The same fused multiply-add, the same order: identical bits.
    if constexpr (DOWN <= 3) {
#pragma unroll
        for (int i = 0; i < NIN; ++i)          // ascending input index; output j meets it with tap NT - 1 + j DOWN - i
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int ti = NT - 1 + j * DOWN - i;
                if (ti >= 0 && ti < NT) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "s"(taps[ti]), "v"(xin[i]));   // <= 64 taps: they all fit the scalar registers
            }
    } else {
        // 85 / 106 / 127 taps: through DPP from six to eight registers (round 4 left down = 4, 5 to the compiler's packed pairs + register moves and had no
        // instance for down = 6: 96 kHz went to the LDS-staged kernel at 12 % of the HBM roofline)
        static_for<0, NIN>([&](auto ic) {
            static_for<0, R>([&](auto jc) {
                constexpr int i = decltype(ic)::value, j = decltype(jc)::value, ti = NT - 1 + j * DOWN - i;
                if constexpr (ti >= 0 && ti < NT) fmac_bcast<ti % 16>(acc[j], tq[ti / 16], xin[i]);
            });
        });
    }
    float4 *dst = reinterpret_cast<float4 *>(y + m0);   // m0 = 10 (mod 4) + multiple of 8: 8-byte aligned only -> two-float stores
    (void)dst;
    if (!mine) return;
#pragma unroll
    for (int j = 0; j < R; j += 2) *reinterpret_cast<float2 *>(y + m0 + j) = make_float2(acc[j], acc[j + 1]);
}

// ------------------------------------------------------------------------------ small interpolation factors (8 / 12 / 24 kHz -> 16 kHz)
// poly_interp_kernel<UP, DOWN, NT>: the register-tiled form of poly_decim_kernel for UP = 2 .. 4.  A thread produces R UP consecutive outputs
// whose first one starts a phase cycle ((m0 + pre_remove) DOWN = 0 mod UP: the host picks the first output accordingly), so the tap index of
// every (output, input) pair is a compile-time constant: taps through the scalar cache as scalar operands of the fused multiply-adds, the
// R DOWN + NT / UP inputs of the thread in registers, no LDS.  (The LDS-staged kernel reads two LDS operands per multiply-add: 11 % of the HBM
// roofline at 8 -> 16 kHz.)  Per output the terms are added in ascending input order over exactly the k range of poly_kernel: identical bits.
// Loads and stores are 16-byte accesses of 4-byte alignment (neither the first input nor the first output of a thread is 16-byte aligned in
// general; the hardware's unaligned access mode serves them).
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
template <int UP, int DOWN, int NT>
__global__ __launch_bounds__(kThreads) void poly_interp_kernel(const float *__restrict__ x, const float *__restrict__ h, float *__restrict__ y, const int64_t m_begin,
                                                               const int64_t q_begin, const int64_t groups) {
    constexpr int R = 4, NO = R * UP, KB = (NT - 1) / UP, NIN = ((NO - 1) * DOWN) / UP + KB + 1, NV = (NIN + 3) / 4;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (g >= groups) return;
    const f4u *src = reinterpret_cast<const f4u *>(x + (q_begin - KB + g * (R * DOWN)));
    float xin[4 * NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { const f4u q = src[v]; xin[4 * v] = q.x; xin[4 * v + 1] = q.y; xin[4 * v + 2] = q.z; xin[4 * v + 3] = q.w; }
    typedef const float __attribute__((address_space(4))) *c_f32;
    const c_f32 taps = (c_f32)h;
    float acc[NO];
#pragma unroll
    for (int r = 0; r < NO; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < NIN; ++i)          // ascending input index; output r meets input i with tap r DOWN + (KB - i) UP
#pragma unroll
        for (int r = 0; r < NO; ++r) {
            const int ti = r * DOWN + (KB - i) * UP;
            if (ti >= 0 && ti < NT) acc[r] = fmaf(taps[ti], xin[i], acc[r]);
        }
    f4u *dst = reinterpret_cast<f4u *>(y + m_begin + g * NO);
#pragma unroll
    for (int v = 0; v < NO / 4; ++v) { f4u o; o.x = acc[4 * v]; o.y = acc[4 * v + 1]; o.z = acc[4 * v + 2]; o.w = acc[4 * v + 3]; dst[v] = o; }
}

// launches poly_interp_kernel on the outputs whose inputs all exist; returns the covered range [m_lo, m_hi) (empty when the pair has no instance)
template <int UP, int DOWN, int NT>
void poly_interp_launch(fa_ctx *ctx, const float *d_x, const float *d_h, float *d_y, int64_t frames, int64_t n_out, int64_t pre_remove, int64_t &m_lo, int64_t &m_hi) {
    constexpr int R = 4, NO = R * UP, KB = (NT - 1) / UP, NIN = ((NO - 1) * DOWN) / UP + KB + 1, NV = (NIN + 3) / 4;
    int64_t m_begin = 0, q_begin = 0, groups = 0;
    fa::interp_geometry(UP, DOWN, NT, R, frames, n_out, pre_remove, m_begin, q_begin, groups);   // resample_geom.h
    (void)KB; (void)NIN; (void)NV; (void)NO;
    m_lo = m_hi = 0;
    if (groups <= 0) return;
    hipLaunchKernelGGL((poly_interp_kernel<UP, DOWN, NT>), dim3(static_cast<unsigned>((groups + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_begin,
                       q_begin, groups);
    m_lo = m_begin; m_hi = m_begin + groups * NO;
}

// ------------------------------------------------------------------------------ non-integer ratios (44.1 / 22.05 / 11.025 kHz -> 16 kHz)
// poly_rows_kernel (round 4).  Output m uses the taps h[p - k up] of its PHASE p mod up, p = (m + pre_remove) down; outputs m and m + up
// share a phase and their input windows lie exactly `down` samples apart.  So lane l of a wavefront takes the outputs m0 + phase + up l:
//   * the taps of a phase are the same in all 64 lanes.  They sit in ONE vector register — lane j holds tap j, lanes 62 / 63 the window offset
//     and the tap count of the phase — fetched by one coalesced 256-byte load a whole chunk of phases ahead, and enter the fused multiply-adds
//     as scalar operands through v_readlane: no LDS read and no memory wait per tap.  (First version: the taps of a phase through the scalar
//     cache, five s_load + a wait in front of every phase — the 37 KB table does not fit the 16 KB scalar cache: 24 % of the HBM roofline.)
//   * the input window of lane l is row l of an LDS tile whose rows are `down` samples apart in the signal and `sld` floats apart in LDS,
//     sld = 4 x odd: every lane's 16-byte reads are aligned and the 64 lanes of a read fall on distinct bank groups.  A window is read from the
//     16-byte boundary below its start; its misalignment (the same in all lanes) is absorbed by the table row, whose taps are shifted by it:
//     NV ds_read_b128 per phase and wavefront instead of two scalar-width LDS operands per multiply-add (poly_lds_kernel: 6.9 % of the HBM
//     roofline at 44.1 -> 16 kHz, LDS-issue bound);
//   * a workgroup = one tile (64 up consecutive outputs) x one group of phases (the phases are split so that the rows of a group fit 74 KB = two
//     workgroups per CU, or 38 KB = three to four for the short windows of many-phase pairs: one stages while the others compute); a wavefront takes CHUNKS of four consecutive phases, so a lane ends a
//     chunk with four consecutive outputs and stores them as one 16-byte piece (the first version stored 4 bytes per lane and phase: one L2
//     transaction per output sample, 128 G transactions per second).
// Summation order per output: ascending input index over the k range of poly_kernel, zero taps in front and behind -> identical bits on finite input.
// The tile geometry (first output, first input, per-phase window offsets and counts) is computed once per rate pair on the host
// (PolyRows below); outputs whose windows touch the ends of the signal go to poly_kernel.
// (PolyRowsGeom, kRowsThreads / kRowsWaves / kRowsOffLane and the host-side geometry: resample_geom.h — shared with the CPU emulation of the tests)
// one phase: `trow` = its table row (one value per lane), rowp = this lane's LDS row (shifted by the group's first staged offset).
// The table row holds the phase's taps SHIFTED by the misalignment of its window (zeros in front and behind), so the window is read from the
// 16-byte boundary below it and every multiply-add has compile-time register indices — no per-alignment code paths.  The padding taps are
// zeros: on finite inputs x * 0 adds +-0 and the sum keeps the bits of poly_kernel's (scipy's upfirdn pads its phases with zeros the same way).
// Round 5 (44.1 kHz: 29 -> see profiles/r05_resample_probe.json): two changes to the inner loop, none to the tiling.
//   * `SHARE` consecutive phases read ONE register window (resample_geom.h): at 44.1 -> 16 kHz two phases share 16 reads (round 4: 16 each; LDS reads
//     of a launch halve), at 22.05 kHz four phases share 10 (round 4: 8 each);
//   * a tap reaches its multiply-add through DPP: v_fmac_f32_dpp ... row_newbcast:n multiplies with lane n of the lane's own 16-lane row of the
//     table register, so a table register holds 16 taps, replicated in its four rows, and a multiply-add is ONE instruction (round 4: v_readlane
//     into a scalar register + v_fmac, two instructions per tap and 62 taps per output: the kernel's arithmetic half).
// Per output the terms are still added in ascending input order over the window, zeros in front and behind: identical bits on finite input.
// (Non-finite input: a zero tap times Inf / NaN is NaN — an Inf or NaN sample reaches every output whose SHARED window covers it, a few samples
// more on either side than its true FIR window; poly_kernel, the edges and the register-tiled kernels skip taps instead.  Documented, tested.)
template <int NTW>
__device__ __forceinline__ void rows_dot(float &acc, const float *tq, const float *xr) {   // acc += sum over the window positions i < NTW of tap i * xr[i], ascending i
#define FA_ROWS_Q(Q)                                                                                                  \
    if constexpr (16 * Q < NTW) {                                                                                     \
        if constexpr (16 * Q + 0 < NTW) fmac_bcast<0>(acc, tq[Q], xr[16 * Q + 0]);   if constexpr (16 * Q + 1 < NTW) fmac_bcast<1>(acc, tq[Q], xr[16 * Q + 1]);   \
        if constexpr (16 * Q + 2 < NTW) fmac_bcast<2>(acc, tq[Q], xr[16 * Q + 2]);   if constexpr (16 * Q + 3 < NTW) fmac_bcast<3>(acc, tq[Q], xr[16 * Q + 3]);   \
        if constexpr (16 * Q + 4 < NTW) fmac_bcast<4>(acc, tq[Q], xr[16 * Q + 4]);   if constexpr (16 * Q + 5 < NTW) fmac_bcast<5>(acc, tq[Q], xr[16 * Q + 5]);   \
        if constexpr (16 * Q + 6 < NTW) fmac_bcast<6>(acc, tq[Q], xr[16 * Q + 6]);   if constexpr (16 * Q + 7 < NTW) fmac_bcast<7>(acc, tq[Q], xr[16 * Q + 7]);   \
        if constexpr (16 * Q + 8 < NTW) fmac_bcast<8>(acc, tq[Q], xr[16 * Q + 8]);   if constexpr (16 * Q + 9 < NTW) fmac_bcast<9>(acc, tq[Q], xr[16 * Q + 9]);   \
        if constexpr (16 * Q + 10 < NTW) fmac_bcast<10>(acc, tq[Q], xr[16 * Q + 10]); if constexpr (16 * Q + 11 < NTW) fmac_bcast<11>(acc, tq[Q], xr[16 * Q + 11]); \
        if constexpr (16 * Q + 12 < NTW) fmac_bcast<12>(acc, tq[Q], xr[16 * Q + 12]); if constexpr (16 * Q + 13 < NTW) fmac_bcast<13>(acc, tq[Q], xr[16 * Q + 13]); \
        if constexpr (16 * Q + 14 < NTW) fmac_bcast<14>(acc, tq[Q], xr[16 * Q + 14]); if constexpr (16 * Q + 15 < NTW) fmac_bcast<15>(acc, tq[Q], xr[16 * Q + 15]); \
    }
    FA_ROWS_Q(0) FA_ROWS_Q(1) FA_ROWS_Q(2) FA_ROWS_Q(3)
#undef FA_ROWS_Q
}

#ifndef FA_ROWS_ATTR
#define FA_ROWS_ATTR   // measurements: -DFA_ROWS_ATTR='__attribute__((amdgpu_waves_per_eu(6, 6)))' caps the registers for three workgroups per CU
#endif
template <int NV, int IT, int SHARE, int HALVES = 1>   // NV: 16-byte reads per window; IT: 64-float pieces per staged row (sld <= 64 IT); SHARE: phases per window;
                                                      // HALVES = 2 (NV = 16, SHARE = 1): a phase of more than 64 taps reads two windows, one right behind the other
__global__ __launch_bounds__(kRowsThreads) FA_ROWS_ATTR void poly_rows_kernel(const float *__restrict__ x, const float *__restrict__ tt, const int2 *__restrict__ gtab,
                                                                float *__restrict__ y, const PolyRowsGeom g, const int64_t tiles, const int64_t m_end, const int vec_ok) {
    extern __shared__ float xs[];
    typedef const int __attribute__((address_space(4))) *c_i32;
    static_assert(HALVES == 1 || (NV == 16 && SHARE == 1), "two windows per phase: full windows, no sharing");
    constexpr int NTW = 4 * NV;                                          // window positions that can carry a tap
    constexpr int NQ = HALVES * ((NTW + 15) / 16);                       // table registers per phase (16 taps each)
    constexpr int NW = 4 / SHARE;                                        // shared windows per chunk of four phases
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt to the 8 XCDs round-robin by their index, and each XCD has its own L2.  The phase groups of ONE tile stage overlapping
    // input spans (a window reaches ~60 samples into the neighbouring group's span: 27 % of the input at 44.1 -> 16 kHz), so they are given indices
    // that differ by a multiple of 8 — the same XCD, dispatched back to back — and the second group finds the overlap in that L2.  (First
    // version: consecutive indices = different XCDs: 896 MB fetched for 635 MB of input, profiles/r04_resample_44100_pmc.json of call 7.)
    const int64_t q = blockIdx.x >> 3;
    const int grp = static_cast<int>(q % g.groups);
    const int64_t tile = (q / g.groups) * 8 + (blockIdx.x & 7);
    if (tile >= tiles) return;
    const int ph0 = grp * g.ppg, ph1 = ph0 + g.ppg < g.up ? ph0 + g.ppg : g.up;
    const int nchunks = (ph1 - ph0 + 3) >> 2;
    const int l16 = lane & 15;
    struct Chunk { float t[4][NQ]; int off[NW]; };
    const int ph_last = g.up - 1;
    auto fetch = [&](Chunk &c, const int p) {                            // table rows + window offsets of the chunk of four phases from p on (p: wave-uniform)
        // requested UNCONDITIONALLY from a clamped phase (a request under a condition is an exec save / restore and a branch around every load): the
        // row of a phase beyond the group's last is never used — its window is skipped or its output is not stored
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t ph = static_cast<size_t>(p + u < ph_last ? p + u : ph_last);
#pragma unroll
            for (int qv = 0; qv < NQ; ++qv) c.t[u][qv] = tt[ph * fa::kRowsTT + 16 * qv + l16];
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const size_t ph = static_cast<size_t>(p + w * SHARE < ph_last ? p + w * SHARE : ph_last);
            c.off[w] = __float_as_int(tt[ph * fa::kRowsTT + fa::kRowsOffPos]);
        }
    };
    int c = wave;
    Chunk cur;
    fetch(cur, ph0 + 4 * c);                                              // requested before the staging
    const c_i32 gt = (c_i32) reinterpret_cast<const int *>(gtab);
    const int smin = gt[2 * grp], span = gt[2 * grp + 1];
    const int64_t kt = g.k_begin + tile * 64 * g.down + smin;
    {   // wavefront w stages rows 8 w .. 8 w + 7 with coalesced 256-byte requests — ALL of them requested before the first is written to LDS
        // (writing each element as it arrived made 40 dependent HBM round trips per thread: 18.8 % of the HBM roofline at 44.1 -> 16 kHz)
        // Round 5: 16 bytes per lane and request (rows start `down` samples apart, i.e. at any 4-byte alignment: the hardware's unaligned access mode
        // serves them, as in poly_interp_kernel; the LDS side is 16-byte aligned: sld and the staged offsets are multiples of 4).  The span of a group
        // is a multiple of 4, so a lane's piece lies inside it or outside it as a whole: 2 requests + 2 LDS writes per row where round 4 issued 5 + 5,
        // each under its own branch.
        constexpr int kRowsPerWave = 64 / kRowsWaves, IT4 = (64 * IT + 255) / 256;
        f4u v[kRowsPerWave][IT4];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const float *src = x + kt + static_cast<int64_t>(wave * kRowsPerWave + r) * g.down;
#pragma unroll
            for (int it = 0; it < IT4; ++it) {
                const int sidx = 4 * lane + 256 * it;
                if (sidx < span) v[r][it] = *reinterpret_cast<const f4u *>(src + sidx);
            }
        }
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            float *dst = xs + (wave * kRowsPerWave + r) * g.sld;
#pragma unroll
            for (int it = 0; it < IT4; ++it) {
                const int sidx = 4 * lane + 256 * it;
                if (sidx < span) *reinterpret_cast<float4 *>(dst + sidx) = make_float4(v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w);
            }
        }
    }
    __syncthreads();
    const float *rowp = xs + lane * g.sld - smin;
    const int64_t mlane = g.m_begin + tile * 64 * g.up + static_cast<int64_t>(lane) * g.up;
    for (; c < nchunks; c += kRowsWaves) {
        const int p = ph0 + 4 * c;
        Chunk nxt;
        fetch(nxt, p + 4 * kRowsWaves);                                   // the next chunk's table rows travel under this chunk's arithmetic
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (p + w * SHARE >= ph1) break;                              // (wave-uniform)
            const int off4 = __builtin_amdgcn_readfirstlane(cur.off[w]);  // the window's first staged offset: a multiple of 4, the same in every lane
            const float4 *wp = reinterpret_cast<const float4 *>(rowp + off4);
            float xr[4 * NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) { const float4 qq = wp[v]; xr[4 * v] = qq.x; xr[4 * v + 1] = qq.y; xr[4 * v + 2] = qq.z; xr[4 * v + 3] = qq.w; }
#pragma unroll
            for (int u = 0; u < SHARE; ++u) rows_dot<NTW>(acc[w * SHARE + u], cur.t[w * SHARE + u], xr);   // (a phase beyond ph1: its output is not stored)
            if constexpr (HALVES == 2) {                                  // the second window of the (single) phase, behind the first: positions 64 .. 127
#pragma unroll
                for (int v = 0; v < NV; ++v) { const float4 qq = wp[NV + v]; xr[4 * v] = qq.x; xr[4 * v + 1] = qq.y; xr[4 * v + 2] = qq.z; xr[4 * v + 3] = qq.w; }
                rows_dot<NTW>(acc[w], cur.t[w] + NQ / 2, xr);
            }
        }
        const int64_t m = mlane + p;
        if (vec_ok && p + 3 < ph1 && m + 3 < m_end) {
            *reinterpret_cast<float4 *>(y + m) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (p + u < ph1 && m + u < m_end) y[m + u] = acc[u];
        }
        cur = nxt;
    }
}

// poly_rows_wide_kernel (round 5): the tile arithmetic of poly_rows_kernel in a PERSISTENT workgroup, one per CU, with two LDS buffers and ALL the phases of
// a tile in one item.  What the measurements of poly_rows_kernel and of two intermediate builds said (profiles/r05_resample_wide_steps.json):
//  * a phase group's staged span is its inputs + one FIR reach (~64 floats): groups of 80 phases stage 1.3 x the signal, groups of 32 phases (what fits two
//    buffers of 64-row tiles) 1.7 x — and a global -> LDS stream alone runs at ~25 GB/s per CU (5.9 TB/s over the chip, the rate MI355X_MICROARCH.md quotes
//    for this path): staging volume, not HBM, set the floor.  Here a tile is 32 rows and holds the whole span of every phase: 1.13 x the signal.
//  * the lanes of a wavefront are 2 phase halves x 32 rows: lane 32 h + r computes row r for the phases 8 c + 4 h .. + 3 of a UNIT c of eight phases (the DPP
//    tap broadcast works within 16-lane rows, so the halves carry different taps).  Unit c = wavefront + 8 j: a wavefront's units — and with them its table
//    rows, CH x 4 x NQ registers — are the same for every tile: they are loaded once per kernel, and a period has no table fetch at all.
//  * the rows of tile i + 1 travel global -> LDS without registers (global_load_lds, 16 bytes per lane: lane l's piece lands at the wave-uniform LDS address
//    + 16 l while every lane brings its own global address, so the [32][sld] image is ONE linear run of pieces) into the buffer tile i - 1 used, while tile i
//    is computed from the other one.  TWO STATIC arrays and a loop unrolled by two, not one array and a toggling offset: the compiler orders LDS reads behind
//    outstanding global -> LDS requests it cannot tell apart from them (s_waitcnt vmcnt(0) in front of the first ds_read: the overlap gone).
//  * a tile's outputs are stored at the START of the next period, in front of that period's requests: the one wait per period (vmcnt(0) + barrier) then covers
//    stores that are a whole period old (their acknowledgement takes ~1 us) and the young requests it is there for.  The barrier is the bare instruction:
//    __syncthreads() carries a workgroup fence, which next to global -> LDS requests the compiler implements as one more s_waitcnt vmcnt(0).
// Same tables, same order of additions: the bits of poly_rows_kernel.
// ROWS = rows of a tile: 32 (one workgroup per CU, two buffers of 64 KB, lanes = 2 phase quads x 32 rows) or 16 (two workgroups per CU, buffers of 32 KB,
// lanes = 4 phase quads x 16 rows — one quad per DPP row); a UNIT is the 4 x 64 / ROWS phases one wavefront-instruction covers.
// floats per LDS row, at most: two buffers of 32 x 512 floats = 128 KB (one workgroup per CU), of 16 x 512 = 64 KB or of 32 x 288 = 72 KB (two per CU);
// 16 x 576 for windows of 32 reads: 88.2 -> 16 kHz stages 564 floats per row
constexpr int wide_row_floats(int rows, int waves, int nv) { return rows == 32 ? (waves == 10 ? (nv == 32 ? 576 : 288) : 512) : (nv == 32 ? 576 : 512); }
template <int ROWS, int WAVES, int NV, int SHARE, int CH>
__device__ __forceinline__ void poly_rows_wide_body(const float *__restrict__ x, const float *__restrict__ tt, float *__restrict__ y, const PolyRowsGeom g_,
                                                    const int2 *__restrict__ gtab, const int tiles_, const int64_t m_end_, const int64_t k_lim_, const int vec_ok_, const int dbg_, const int rot) {
    constexpr int kWideBuf = ROWS * wide_row_floats(ROWS, WAVES, NV), PU = 4 * 64 / ROWS;     // floats per LDS buffer; phases per unit
    __shared__ float buf_a[kWideBuf];
    __shared__ float buf_b[kWideBuf];
    constexpr int NTW = 4 * NV, NQ = (NTW + 15) / 16, NW = 4 / SHARE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l16 = lane & 15, quad = lane / ROWS;
    // the arguments the loop uses, as values the compiler cannot re-load: under register pressure it would fetch kernel arguments again INSIDE the loop, and a
    // scalar load outstanding next to LDS reads leaves it only s_waitcnt lgkmcnt(0) (scalar loads return out of order) — the read pipeline below gone
    int64_t m_end = m_end_, k_lim = k_lim_, k_begin = g_.k_begin, m_begin = g_.m_begin;
    // a workgroup serves ONE phase group for the whole kernel (its wavefronts' table rows never change) and every n_chains-th tile: workgroup b -> XCD b & 7,
    // group (b >> 3) % groups, tiles (b & 7) + 8 ((b >> 3) / groups) + k n_chains — the workgroups that stage the overlapping spans of a tile's groups are 8
    // apart: the same XCD's L2, at about the same time
    const int groups = g_.groups, qb = static_cast<int>(blockIdx.x) >> 3, grp = qb % groups;
    int up = g_.up, down = g_.down, sld = g_.sld, tiles = tiles_, vec_ok = vec_ok_, dbg = dbg_, smin = reinterpret_cast<const int *>(gtab)[2 * grp];
    int n_chains = 8 * (static_cast<int>(gridDim.x) / (8 * groups)), ph0 = grp * g_.ppg, ph1 = ph0 + g_.ppg < up ? ph0 + g_.ppg : up;
    int tile = (static_cast<int>(blockIdx.x) & 7) + 8 * (qb / groups);
    asm volatile("" : "+s"(m_end), "+s"(k_lim), "+s"(k_begin), "+s"(m_begin), "+s"(up), "+s"(down), "+s"(sld), "+s"(tiles), "+s"(vec_ok), "+s"(dbg), "+s"(smin), "+s"(n_chains),
                 "+s"(ph0), "+s"(ph1), "+s"(tile));
    struct { int64_t k_begin, m_begin; int up, down, sld; } g{k_begin, m_begin, up, down, sld};
    const int n_units = (ph1 - ph0 + PU - 1) / PU;
    if (qb / groups >= static_cast<int>(gridDim.x) / (8 * groups)) return;      // (a grid that is not a multiple of 8 groups: the workgroups behind the last whole set)
    // this wavefront's units: table rows (the table has eight empty rows behind the last phase) and LDS window addresses, for the whole kernel
    float t[CH][4][NQ];
    int xa[CH][NW], rj[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int unit = wave + WAVES * j < n_units ? wave + WAVES * j : n_units - 1;
        const float *row = tt + static_cast<size_t>(ph0 + PU * unit + 4 * quad) * fa::kRowsTT;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int qv = 0; qv < NQ; ++qv) t[j][u][qv] = row[u * fa::kRowsTT + 16 * qv + l16];
        int woff[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) woff[w] = __float_as_int(row[w * SHARE * fa::kRowsTT + fa::kRowsOffPos]) - smin;
        // the row of a lane.  ROWS = 32: lane % 32.  ROWS = 16: a ds_read_b128 is served in groups of 16 lanes that belong to TWO phase quads, whose windows start
        // at different offsets — 16-byte slot (row sld / 4 + offset / 4) mod 16: the rows of a group are all different (sld / 4 is odd), the two offsets shift
        // eight of them onto the other eight's slots, 2-way conflicts on every read (profiles/r05_resample_wide_w16_8_dbg0_pmc.json: half of the LDS cycles).  A quad
        // therefore takes its rows ROTATED by rot (-offset / 4) with rot = (sld / 4)^-1 mod 16: lane i of every quad then reads slot i sld / 4 of the unit's first
        // window — conflict-free; a unit's second window (two phases further: 1 or 2 slots, not always the same for two quads) keeps some
        if constexpr (ROWS == 16) rj[j] = ((lane & 15) + rot * (16 - ((woff[0] >> 2) & 15))) & 15; else rj[j] = lane % ROWS;
#pragma unroll
        for (int w = 0; w < NW; ++w) xa[j][w] = 4 * (rj[j] * g.sld + woff[w]);     // (bytes)
    }
    // staging: piece i (16 bytes) of the linear LDS image = row i / (sld / 4), floats 4 (i % (sld / 4)) .. + 3 of that row; the pieces behind the 32nd row
    // (8 sld < 4096) re-read the last one into the buffer's unused tail — every wavefront issues the same eight requests, no branches.  Rows start at odd
    // multiples of 4 bytes: unaligned 16-byte requests, as in poly_rows_kernel.  The floats behind a row's span are never read by the arithmetic; where they
    // would lie behind the signal's end the address is clamped (k_lim).
    constexpr int kThreadsW = 64 * WAVES, kPieces = (kWideBuf / 4 + kThreadsW - 1) / kThreadsW;
    int soff[kPieces];
    const int sld4 = g.sld / 4;
#pragma unroll
    for (int it = 0; it < kPieces; ++it) {
        int i = it * kThreadsW + static_cast<int>(threadIdx.x);
        if (i >= ROWS * sld4) i = ROWS * sld4 - 1;
        const int row = i / sld4, c4 = i - row * sld4;
        soff[it] = row * g.down + 4 * c4;
    }
    const int tile_step = ROWS * g.down;
    auto stage = [&](const int tile, float *buf) {
        const int64_t kt = g.k_begin + static_cast<int64_t>(tile) * tile_step + smin;
        const float *src = x + kt;
        const int64_t lim64 = k_lim - kt;
        const int lim = lim64 < 0x3fffffff ? static_cast<int>(lim64) : 0x3fffffff;
#pragma unroll
        for (int it = 0; it < kPieces; ++it) {
            const int base = it * kThreadsW + wave * 64;         // first piece of this wavefront's request
            if (kWideBuf / 4 % kThreadsW == 0 || base < kWideBuf / 4) __builtin_amdgcn_global_load_lds(src + (soff[it] < lim ? soff[it] : lim), buf + 4 * base, 16, 0, 2);
        }
    };
    if (tile >= tiles) return;
    stage(tile, buf_a);
    // (the table registers pass through an empty statement: the compiler waits for their loads HERE, not at their first use inside the loop — behind the
    // next tile's requests, which it would wait out with them)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int qv = 0; qv < NQ; ++qv) asm volatile("" : "+v"(t[j][u][qv]));
#pragma unroll
        for (int w = 0; w < NW; ++w) asm volatile("" : "+v"(xa[j][w]));
        asm volatile("" : "+v"(rj[j]));
    }
    float st_acc[CH][4];
    int st_tile = -1;
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int AHEAD = (WAVES == 10 && NV != 32) ? 1 : (ROWS == 32 ? 3 : 2), RING = AHEAD + 1;    // pieces read ahead (two workgroups per CU: 4 - 5 wavefronts per SIMD cover each other, and 128 / 96 registers are all there is; the long-window ten-wavefront instance is ONE workgroup per CU with 170 registers: three ahead, 0.530 -> 0.512 ms at 88.2 kHz)
    f4v xq[RING][4] = {};                                   // the ring of window quarters (see the arithmetic below)
    auto flush = [&]() {
        if (st_tile < 0) return;
        const int64_t m_tile = g.m_begin + static_cast<int64_t>(st_tile) * ROWS * g.up;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (wave + WAVES * j >= n_units) break;
            const int ph = ph0 + PU * (wave + WAVES * j) + 4 * quad;
            const int64_t m = m_tile + rj[j] * g.up + ph;
            if (vec_ok && ph + 3 < ph1 && m + 3 < m_end) {
                f4v o = {st_acc[j][0], st_acc[j][1], st_acc[j][2], st_acc[j][3]};
                *reinterpret_cast<f4v *>(y + m) = o;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (ph + u < ph1 && m + u < m_end) y[m + u] = st_acc[j][u];
            }
        }
        st_tile = -1;
    };
    // one tile: computed from `mine` while the next one travels into `other`
    auto step = [&](const float *mine, float *other) {
        const int tile_n = tile + n_chains;
        // (dbg = FA_RESAMPLE_WIDE_PART, 0 in production: 1 = staging only, 2 = everything but staging, 3 = arithmetic and stores without the period's wait and
        // barrier, 4 = arithmetic alone — the decomposition of profiles/r05_resample_wide_steps.json; wave-uniform branches on a scalar)
        if (dbg < 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's pieces of the present tile have landed ...
            __builtin_amdgcn_s_barrier();                       // ... everybody's have, and nobody reads the other buffer any more
            asm volatile("" ::: "memory");
        }
        if (dbg < 4) flush();
        if (tile_n < tiles && dbg < 2) stage(tile_n, other);
        if (dbg != 1) {
            // a software pipeline over PIECES of a window (four 16-byte LDS reads = 16 window positions): the reads of piece i + AHEAD are issued before the
            // multiply-adds of piece i, through a ring of AHEAD + 1 register quarters.  A CU holds ONE workgroup here, two wavefronts per SIMD: nobody else covers a
            // read's latency (first build, whole windows read and then used: 4.5 us of arithmetic per tile where the multiply-adds need 2.4 —
            // profiles/r05_resample_wide_steps.json).  At most three pieces ahead = 12 reads in flight: the LDS counter (lgkmcnt) has four bits.
            // The reads and their waits are WRITTEN OUT (ds_read_b128 / s_waitcnt lgkmcnt(n) statements): left to the compiler, every variant of this loop
            // ended in s_waitcnt lgkmcnt(0) in front of each piece (the wavefront-uniform branch around a unit's arithmetic, a kernel argument re-loaded inside
            // the loop, the scheduler moving a later window's reads first) or in s_waitcnt vmcnt(0) in front of the first read (the next tile's requests).
            // A wait names the reads issued behind the piece it is for; the counter retires LDS reads in issue order, and a scalar load the compiler might
            // add in between only makes the wait longer (it raises the outstanding count, never lowers it below what the reads in front need).  The
            // multiply-adds take their inputs from the wait statement's outputs: they cannot move in front of it.
            constexpr int PC = (NV + 3) / 4, NP = CH * NW * PC;
            const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) float *) mine));
            auto reads_of = [](const int i) { const int pc = i % PC; return NV - 4 * pc < 4 ? NV - 4 * pc : 4; };
            auto load = [&](auto ic) {
                constexpr int i = decltype(ic)::value, j = i / (NW * PC), w = (i / PC) % NW, pc = i % PC;
                const unsigned addr = lds_base + static_cast<unsigned>(xa[j][w]);
                f4v *q = xq[i % RING];
                if constexpr (4 * pc + 0 < NV) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[0]) : "v"(addr), "n"(64 * pc));
                if constexpr (4 * pc + 1 < NV) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[1]) : "v"(addr), "n"(64 * pc + 16));
                if constexpr (4 * pc + 2 < NV) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[2]) : "v"(addr), "n"(64 * pc + 32));
                if constexpr (4 * pc + 3 < NV) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[3]) : "v"(addr), "n"(64 * pc + 48));
            };
            static_for<0, (NP < AHEAD ? NP : AHEAD)>(load);
            static_for<0, NP>([&](auto ic) {
                constexpr int i = decltype(ic)::value, j = i / (NW * PC), w = (i / PC) % NW, pc = i % PC;
                // (the reads are unconditional — behind a wavefront's last real unit they fetch pieces nobody uses; skipping them, with waits that name only the
                // reads issued, measured slower where most wavefronts have all their units: 22.05 kHz 125 -> 140 us, 11.025 kHz 117 -> 154 us per audio hour)
                if constexpr (i + AHEAD < NP) load(std::integral_constant<int, i + AHEAD>{});
                constexpr int behind = (i + 1 < NP && AHEAD >= 1 ? reads_of(i + 1) : 0) + (i + 2 < NP && AHEAD >= 2 ? reads_of(i + 2) : 0) + (i + 3 < NP && AHEAD >= 3 ? reads_of(i + 3) : 0);
                f4v *q = xq[i % RING];
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(behind));
                if (wave + WAVES * j < n_units) {               // (wave-uniform)
                    if constexpr (w == 0 && pc == 0) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) st_acc[j][u] = 0.0f;
                    }
                    static_for<0, 16>([&](auto pcst) {          // position by position, the phases of the window side by side (independent accumulators)
                        constexpr int pos = decltype(pcst)::value;
                        if constexpr (16 * pc + pos < NTW) {
#pragma unroll
                            for (int u = 0; u < SHARE; ++u) fmac_bcast<pos>(st_acc[j][w * SHARE + u], t[j][w * SHARE + u][pc], q[pos / 4][pos % 4]);
                        }
                    });
                }
            });
            st_tile = tile;
        }
        tile = tile_n;
    };
    for (;;) {
        step(buf_a, buf_b);
        if (tile >= tiles) break;
        step(buf_b, buf_a);
        if (tile >= tiles) break;
    }
    flush();
}

// (separate kernels, not one with launch bounds that depend on ROWS / WAVES: hipcc emits no host stub for a kernel template whose bounds are value-dependent)
#define FA_WIDE_KERNEL(NAME, ROWS_, WAVES_, MINW)                                                                                                     \
    template <int NV, int SHARE, int CH>                                                                                                              \
    __global__ __launch_bounds__(64 * WAVES_, MINW) void NAME(const float *__restrict__ x, const float *__restrict__ tt, float *__restrict__ y, const PolyRowsGeom g, \
                                                              const int2 *__restrict__ gtab, const int tiles, const int64_t m_end, const int64_t k_lim, const int vec_ok, const int dbg, const int rot) { \
        poly_rows_wide_body<ROWS_, WAVES_, NV, SHARE, CH>(x, tt, y, g, gtab, tiles, m_end, k_lim, vec_ok, dbg, rot);                                        \
    }
FA_WIDE_KERNEL(poly_rows_wide32_kernel, 32, 8, 2)        // one workgroup per CU: 2 wavefronts per SIMD, 256 registers
FA_WIDE_KERNEL(poly_rows_wide16_kernel, 16, 8, 4)        // two per CU: 4 per SIMD, 128 registers
FA_WIDE_KERNEL(poly_rows_wide16w10_kernel, 16, 10, 5)    // two of ten wavefronts: 5 per SIMD, 96 registers
FA_WIDE_KERNEL(poly_rows_wide32w10l_kernel, 32, 10, 3)   // windows of 32 reads (88.2 kHz: rows of 564 floats, two buffers of 72 KB): ONE workgroup of ten wavefronts per CU, 170 registers
FA_WIDE_KERNEL(poly_rows_wide32w10_kernel, 32, 10, 5)    // the same with 32-row tiles of one phase GROUP (rows of <= 288 floats): groups of 80 k phases = 10 k units (44.1 / 22.05 / 11.025 kHz)
#undef FA_WIDE_KERNEL

// ------------------------------------------------------------------------------ integer decimation through LDS tiles (32 / 48 / 64 / 80 / 96 kHz -> 16 kHz)
// poly_decim_tile_kernel<DOWN> (round 5).  poly_decim_kernel keeps a thread's whole window in registers and fetches it with 16-byte loads that lie R DOWN
// floats apart from lane to lane: a wavefront's load touches 48 cache lines (DOWN = 3) where a contiguous one touches 8, and the address unit's line rate,
// not HBM, bounds the kernel (22 loads x 48 lines per wavefront: 193 us of the 270 us per audio hour at 48 kHz; VALU busy 6 %).  Here a persistent workgroup
// copies the CONTIGUOUS input span of a tile of 256 R outputs global -> LDS (global_load_lds_dwordx4: 16 bytes per lane, 8 lines per request) into one of two
// buffers while the previous tile is computed from the other one — the scheme of poly_rows_wide_body: two static arrays, one wait + bare barrier per tile,
// stores at the start of the next period.  Thread i computes the tile's outputs R i .. R i + R - 1 from the LDS floats [R DOWN i, + NIN): R DOWN = 4 x odd, so a
// wavefront's 16-byte LDS reads are conflict-free; the window streams through two register quarters of 16 positions; a tap is the SCALAR operand of its
// v_fmac_f32 (plain multiply-adds issue 1.6 x faster than the DPP form, profiles/r05_dpp_rate_ubench.jsonl), fetched through the scalar cache block by block
// (a block of 16 positions meets 16 + (R - 1) DOWN taps: the pointer passes through an empty statement per block, or the compiler hoists all 127 loads).
// The same fused multiply-adds in the same order as poly_decim_kernel: ascending input index per output.
using fa::DecimTile;   // (resample_geom.h: shared with the CPU replay of the tests)
static_assert(fa::kDecimThreads == kThreads, "one thread per R outputs of a tile");
template <int DOWN>
__global__ __launch_bounds__(kThreads) void poly_decim_tile_kernel(const float *__restrict__ x, const float *__restrict__ h, float *__restrict__ y, const int64_t m_begin_,
                                                                   const int tiles_, const int64_t k_lim_) {
    typedef DecimTile<DOWN> D;
    constexpr int R = D::R, NT = D::NT, RS = D::RS, NIN = D::NIN, NB = D::NB, TO = D::TO;
    __shared__ __attribute__((aligned(16))) float buf_a[D::BUF];
    __shared__ __attribute__((aligned(16))) float buf_b[D::BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int64_t m_begin = m_begin_, k_lim = k_lim_;
    int tiles = tiles_, grid = static_cast<int>(gridDim.x);
    asm volatile("" : "+s"(m_begin), "+s"(k_lim), "+s"(tiles), "+s"(grid));      // (no kernel argument re-loaded inside the loop: see poly_rows_wide_body)
    typedef const float __attribute__((address_space(4))) *c_f32;
    auto stage = [&](const int tile, float *buf) {
        const int64_t k0 = (m_begin + static_cast<int64_t>(tile) * TO - 10) * DOWN;      // first input of the tile (a multiple of 4: m_begin = 10 mod 4)
        const float *src = x + k0;
        const int64_t lim64 = k_lim - k0;
        const int lim = lim64 < 0x3fffffff ? static_cast<int>(lim64) : 0x3fffffff;
        constexpr int NREQ = (D::PIECES + 63) / 64;                          // requests of 64 pieces, dealt round-robin to the four wavefronts
#pragma unroll
        for (int it = 0; it < (NREQ + 3) / 4; ++it) {
            const int req = it * 4 + wave;
            const int off = 4 * (req * 64 + lane);
            if (req < NREQ) __builtin_amdgcn_global_load_lds(src + (off < lim ? off : lim), buf + 256 * req, 16, 0, 2);
        }
    };
    int tile = blockIdx.x;
    if (tile >= tiles) return;
    stage(tile, buf_a);
    float st_acc[R];
    int st_tile = -1;
    auto flush = [&]() {
        if (st_tile < 0) return;
        float *dst = y + m_begin + static_cast<int64_t>(st_tile) * TO + R * tid;      // 8-byte aligned (m_begin = 2 mod 4, R even) or 4-byte (R = 3)
        if constexpr (R == 4) { f4u o; o.x = st_acc[0]; o.y = st_acc[1]; o.z = st_acc[2]; o.w = st_acc[3]; *reinterpret_cast<f4u *>(dst) = o; }
        else if constexpr (R % 2 == 0) {
#pragma unroll
            for (int j = 0; j < R; j += 2) *reinterpret_cast<float2 *>(dst + j) = make_float2(st_acc[j], st_acc[j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) dst[j] = st_acc[j];
        }
        st_tile = -1;
    };
    auto step = [&](const float *mine, float *other) {
        const int tile_n = tile + grid;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's pieces of the present tile have landed ...
        __builtin_amdgcn_s_barrier();                       // ... everybody's have, and nobody reads the other buffer any more
        asm volatile("" ::: "memory");
        flush();
        if (tile_n < tiles) stage(tile_n, other);
        // the LDS reads and their waits are written out (see poly_rows_wide_body: the compiler would wait for the next tile's requests in front of the first
        // read — it cannot know that the wait above already covered this buffer's); a scalar tap load the compiler adds in between only lengthens a wait
        const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) float *) mine)) + 4u * static_cast<unsigned>(tid * RS);
        float acc[R];
#pragma unroll
        for (int j = 0; j < R; ++j) acc[j] = 0.0f;
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v q[2][4] = {};
        auto load = [&](auto bc) {
            constexpr int b = decltype(bc)::value;
            f4v *d = q[b & 1];
            const unsigned a = addr;                                         // (a local of this lambda: hipcc rejects captured variables as asm operands of a generic lambda)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[0]) : "v"(a), "n"(64 * b));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[1]) : "v"(a), "n"(64 * b + 16));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[2]) : "v"(a), "n"(64 * b + 32));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[3]) : "v"(a), "n"(64 * b + 48));
        };
        load(std::integral_constant<int, 0>{});
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if constexpr (b + 1 < NB) load(std::integral_constant<int, b + 1>{});
            f4v *d = q[b & 1];
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "n"(b + 1 < NB ? 4 : 0));
            c_f32 taps = (c_f32) h;
            asm volatile("" : "+s"(taps));                                   // (this block's taps are fetched here, not with everybody else's at the top)
            static_for<0, 16>([&](auto pc) {
                constexpr int p = 16 * b + decltype(pc)::value;              // ascending input index; output j meets it with tap NT - 1 + j DOWN - p
                if constexpr (p < NIN) {
                    const float xv = d[(p % 16) / 4][p % 4];
                    static_for<0, R>([&](auto jc) {
                        constexpr int j = decltype(jc)::value, ti = NT - 1 + j * DOWN - p;
                        if constexpr (ti >= 0 && ti < NT) fmac_scalar(acc[j], taps[ti], xv);
                    });
                }
            });
        });
#pragma unroll
        for (int j = 0; j < R; ++j) st_acc[j] = acc[j];
        st_tile = tile;
        tile = tile_n;
    };
    for (;;) {
        step(buf_a, buf_b);
        if (tile >= tiles) break;
        step(buf_b, buf_a);
        if (tile >= tiles) break;
    }
    flush();
}

// host side of poly_rows_kernel: the tables of one rate pair, resident on the device with the context
struct PolyRows {
    bool wide = false;              // served by poly_rows_wide_kernel (32-row tiles, every phase in one item, two LDS buffers)
    int ch = 0, wide_rows = 0, wide_waves = 0;   // its units per wavefront; rows per tile (32: units of 8 phases, 16: of 16); wavefronts per workgroup
    bool wide_no_rot = false;
    int wide_part = 0;              // FA_RESAMPLE_WIDE_PART (read when the tables are built): time PARTS of the kernel — results are wrong for any value but 0
    PolyRowsGeom g{};
    int nv = 0;                     // 16-byte reads per phase window
    size_t lds = 0;
    void *d_tables = nullptr;       // [gtab int2 x groups, padded to 256 B][table rows: 64 floats per phase]
    size_t tt_offset = 0;
    int up = 0, down = 0;
    ~PolyRows() { if (d_tables) (void)hipFree(d_tables); }
};
void poly_rows_free(void *p) { delete static_cast<PolyRows *>(p); }
bool wide_instance(int rows, int waves, int nv, int share, int ch);

// Geometry + tables (resample_geom.h); false when the pair does not suit the kernel (then poly_lds_kernel serves it).
bool poly_rows_build(PolyRows &R, const std::vector<float> &h, int up, int down, int64_t pre_remove, std::vector<int> &gtab, std::vector<float> &tt) {
    size_t budget = 0;                                                  // automatic (resample_geom.h); FA_RESAMPLE_ROWS_LDS_KB: measurements (r04_rows_lds_probe.json)
    if (const char *e = fa::sw(fa::Sw::RESAMPLE_ROWS_LDS_KB)) { const int v = atoi(e); if (v >= 16 && v <= 150) budget = static_cast<size_t>(v) * 1024; }
    int share_max = 4;                                                  // FA_RESAMPLE_ROWS_SHARE = 1 | 2 | 4: measurements (1 = every phase its own window, round 4's reads)
    if (const char *e = fa::sw(fa::Sw::RESAMPLE_ROWS_SHARE)) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) share_max = v; }
    // the persistent kernel with two buffers (FA_RESAMPLE_NO_WIDE=1: the one-tile-per-workgroup kernel): one phase group — an unbounded LDS budget keeps
    // rows_geometry from splitting —, rows of at most 512 floats, and one of the instantiated (window, sharing, units per wavefront) combinations
    R.wide = false;
    if (budget == 0 && !fa::sw_on(fa::Sw::RESAMPLE_NO_WIDE)) {
        // candidates, in order: {rows, wavefronts, LDS budget of rows_geometry (per 64 rows: it decides the phase groups)}.  FA_RESAMPLE_WIDE = "rows:waves" picks
        // one.  Measured per audio hour (profiles/r05_resample_wide_steps.json): 16-row tiles, 8 wavefronts, two workgroups per CU — 44.1 kHz 233 - 250 us,
        // 22.05 kHz 128, 11.025 kHz 116; 32-row tiles with one workgroup per CU 243 - 262 / 152 / (no instance); 32-row tiles of one phase GROUP (80 k phases:
        // rows of <= 288 floats, two workgroups of ten wavefronts, no LDS bank conflicts, every wavefront the same number of units) 245 - 257 / 152 / 159.
        // The 16-row form wins or ties in spite of its 2-way LDS bank conflicts (two phase quads with different window offsets share a ds_read_b128 lane group)
        struct Cand { int rows, waves; size_t budget; };
        const size_t group_budget = size_t{64} * 288 * 4;      // (rows_geometry budgets 64 rows)
        // (round 6: windows of 32 reads — 88.2 kHz, 80 phases — first try 32-row tiles with ten wavefronts: 10 units of 8 phases, one per wavefront; with 16-row
        // tiles the 5 units of 16 phases leave three of eight wavefronts without work, and those still issue their share of the LDS reads)
        std::vector<Cand> cands = {{32, 10, size_t{1} << 30}, {16, 8, size_t{1} << 30}, {32, 10, group_budget}, {32, 8, size_t{1} << 30}};
        if (const char *e = fa::sw(fa::Sw::RESAMPLE_WIDE)) {
            int r_ = 0, w_ = 0;
            // only the instantiated forms: any other pair would reach the geometry arithmetic below (rows 0: a division by zero)
            if (sscanf(e, "%d:%d", &r_, &w_) == 2 && (r_ == 16 || r_ == 32) && (w_ == 8 || w_ == 10))
                cands = r_ == 32 && w_ == 10 ? std::vector<Cand>{{32, 10, size_t{1} << 30}, {32, 10, group_budget}} : std::vector<Cand>{{r_, w_, size_t{1} << 30}};
        }
        for (const Cand &c : cands) {
            PolyRowsGeom g2{};
            int nv2 = 0;
            std::vector<int> gtab2;
            std::vector<float> tt2;
            if (!fa::rows_geometry(g2, nv2, h, up, down, pre_remove, gtab2, tt2, c.budget, share_max)) continue;
            if (c.rows == 32 && c.waves == 10 && c.budget > group_budget && nv2 != 32) continue;   // the ungrouped ten-wavefront form exists for the long windows only
            if (g2.sld > wide_row_floats(c.rows, c.waves, nv2) || g2.groups > 8) continue;
            if (!(c.rows == 32 && c.waves == 10) && g2.groups != 1) continue;
            const int pu = 4 * 64 / c.rows, units = (g2.ppg + pu - 1) / pu, ch = (units + c.waves - 1) / c.waves;
            if (c.waves == 10 && units % 10 != 0) continue;              // (ten wavefronts only where they divide the units)
            if (!wide_instance(c.rows, c.waves, nv2, g2.share, ch)) continue;
            R.g = g2; R.nv = nv2; R.ch = ch; R.wide_rows = c.rows; R.wide_waves = c.waves; gtab.swap(gtab2); tt.swap(tt2);
            R.wide = true;
            if (const char *e = fa::sw(fa::Sw::RESAMPLE_WIDE_PART)) R.wide_part = atoi(e);
            R.wide_no_rot = fa::sw_on(fa::Sw::RESAMPLE_WIDE_NO_ROT);
            R.lds = 0; R.up = up; R.down = down;     // (static LDS)
            return true;
        }
    }
    if (!fa::rows_geometry(R.g, R.nv, h, up, down, pre_remove, gtab, tt, budget, share_max)) return false;
    if (R.g.sld > 64 * 5 && R.g.share != 1 && !fa::rows_geometry(R.g, R.nv, h, up, down, pre_remove, gtab, tt, budget, 1)) return false;   // the long-row build is instantiated for share = 1 only
    R.lds = static_cast<size_t>(R.g.sld) * 64 * sizeof(float); R.up = up; R.down = down;
    return true;
}

template <int NV, int IT, int SHARE, int HALVES = 1>
void poly_rows_launch_it(fa_ctx *ctx, const PolyRows &R, const float *d_x, float *d_y, int64_t tiles, int64_t m_end) {
    const int2 *gtab = static_cast<const int2 *>(R.d_tables);
    const float *tt = reinterpret_cast<const float *>(static_cast<const char *>(R.d_tables) + R.tt_offset);
    const int vec_ok = R.up % 4 == 0 && (reinterpret_cast<uintptr_t>(d_y) & 15) == 0 ? 1 : 0;   // m_begin and the chunk starts are multiples of 4
    if (R.lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(poly_rows_kernel<NV, IT, SHARE, HALVES>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(R.lds));
    const int64_t tiles8 = (tiles + 7) / 8 * 8;   // whole rounds of the 8 XCDs: the index -> (tile, group) map of the kernel
    hipLaunchKernelGGL((poly_rows_kernel<NV, IT, SHARE, HALVES>), dim3(static_cast<unsigned>(tiles8 * R.g.groups)), dim3(kRowsThreads), R.lds, ctx->stream, d_x, tt, gtab, d_y, R.g, tiles, m_end, vec_ok);
}
// the instantiated combinations of the wide kernels: {rows per tile, wavefronts, 16-byte reads per window, phases per window, units per wavefront}
//   44.1 -> 16 kHz: windows {16, 2}, 10 units of 16 phases or 20 of 8; 22.05 -> 16 kHz {10, 4}, 20 or 40 units; 11.025 -> 16 kHz {8, 4}, 40 units; 37.8 -> 16 kHz {16, 4}, 5 units;
//   88.2 -> 16 kHz {32, 1} (two windows' worth of reads per phase), 5 units; other pairs: poly_rows_kernel
#define FA_WIDE_INSTANCES(X) X(32, 8, 16, 2, 3) X(32, 8, 10, 4, 5) X(16, 8, 16, 2, 2) X(16, 8, 10, 4, 3) X(16, 8, 8, 4, 5) X(16, 8, 16, 4, 1) X(16, 8, 32, 1, 1) X(16, 10, 16, 2, 1) X(16, 10, 10, 4, 2) X(16, 10, 8, 4, 4) \
    X(32, 10, 16, 2, 1) X(32, 10, 10, 4, 2) X(32, 10, 8, 4, 4) X(32, 10, 16, 4, 1) X(32, 10, 32, 1, 1)
bool wide_instance(int rows, int waves, int nv, int share, int ch) {
#define FA_WIDE_IS(R_, W_, V, S, C) if (rows == R_ && waves == W_ && nv == V && share == S && ch == C) return true;
    FA_WIDE_INSTANCES(FA_WIDE_IS)
#undef FA_WIDE_IS
    return false;
}
void poly_rows_wide_launch(fa_ctx *ctx, const PolyRows &R, const float *d_x, float *d_y, int64_t tiles, int64_t m_end, int64_t frames) {
    const int2 *gtab = static_cast<const int2 *>(R.d_tables);
    const float *tt = reinterpret_cast<const float *>(static_cast<const char *>(R.d_tables) + R.tt_offset);
    const int vec_ok = R.up % 4 == 0 && (reinterpret_cast<uintptr_t>(d_y) & 15) == 0 ? 1 : 0;
    // the resident workgroups: one (128 KB of LDS) or two (64 / 72 KB) per CU, in whole sets of 8 x groups (kernel: workgroup -> XCD, group, tile chain)
    const int per_cu = R.wide_rows == 32 && (R.wide_waves == 8 || R.nv == 32) ? 1 : 2, set = 8 * R.g.groups;
    int64_t sets = 256 * per_cu / set;
    sets = std::max<int64_t>(1, std::min<int64_t>(sets, (tiles + 7) / 8));
    const unsigned grid = static_cast<unsigned>(sets * set);
    const int dbg = R.wide_part;
    int rot = 0;                                                     // (sld / 4)^-1 mod 16 (sld / 4 is odd); FA_RESAMPLE_WIDE_NO_ROT (tables' build time): rows unrotated
    for (int c = 1; c < 16; c += 2) if ((c * (R.g.sld / 4)) % 16 == 1) rot = c;
    if (R.wide_no_rot) rot = 0;
    const int n_tiles = static_cast<int>(tiles);
    const int64_t k_lim = frames - 4;
#define FA_WIDE_GO(R_, W_, V, S, C)                                                                                                                 \
    if (R.wide_rows == R_ && R.wide_waves == W_ && R.nv == V && R.g.share == S && R.ch == C) {                                                     \
        if constexpr (R_ == 32 && W_ == 8) hipLaunchKernelGGL((poly_rows_wide32_kernel<V, S, C>), dim3(grid), dim3(64 * W_), 0, ctx->stream, d_x, tt, d_y, R.g, gtab, n_tiles, m_end, k_lim, vec_ok, dbg, rot);   \
        else if constexpr (R_ == 32 && V == 32) hipLaunchKernelGGL((poly_rows_wide32w10l_kernel<V, S, C>), dim3(grid), dim3(64 * W_), 0, ctx->stream, d_x, tt, d_y, R.g, gtab, n_tiles, m_end, k_lim, vec_ok, dbg, rot); \
        else if constexpr (R_ == 32) hipLaunchKernelGGL((poly_rows_wide32w10_kernel<V, S, C>), dim3(grid), dim3(64 * W_), 0, ctx->stream, d_x, tt, d_y, R.g, gtab, n_tiles, m_end, k_lim, vec_ok, dbg, rot); \
        else if constexpr (W_ == 8) hipLaunchKernelGGL((poly_rows_wide16_kernel<V, S, C>), dim3(grid), dim3(64 * W_), 0, ctx->stream, d_x, tt, d_y, R.g, gtab, n_tiles, m_end, k_lim, vec_ok, dbg, rot); \
        else hipLaunchKernelGGL((poly_rows_wide16w10_kernel<V, S, C>), dim3(grid), dim3(64 * W_), 0, ctx->stream, d_x, tt, d_y, R.g, gtab, n_tiles, m_end, k_lim, vec_ok, dbg, rot); \
        return;                                                                                                                                    \
    }
    FA_WIDE_INSTANCES(FA_WIDE_GO)
#undef FA_WIDE_GO
}
template <int NV>
void poly_rows_launch(fa_ctx *ctx, const PolyRows &R, const float *d_x, float *d_y, int64_t tiles, int64_t m_end, int64_t frames) {
    if constexpr (NV == 32) {                                                               // phases of 65 .. 128 taps (88.2 -> 16 kHz): two windows of 16 reads per phase
        if (R.g.sld > 64 * 5) poly_rows_launch_it<16, 10, 1, 2>(ctx, R, d_x, d_y, tiles, m_end);
        else poly_rows_launch_it<16, 5, 1, 2>(ctx, R, d_x, d_y, tiles, m_end);
    } else
    if (R.g.sld > 64 * 5) poly_rows_launch_it<NV, 10, 1>(ctx, R, d_x, d_y, tiles, m_end);   // few phases with long windows: one group per tile (built with share = 1)
    else if (R.g.share == 4) poly_rows_launch_it<NV, 5, 4>(ctx, R, d_x, d_y, tiles, m_end);  // rows of a group within 74 KB (the common case)
    else if (R.g.share == 2) poly_rows_launch_it<NV, 5, 2>(ctx, R, d_x, d_y, tiles, m_end);
    else poly_rows_launch_it<NV, 5, 1>(ctx, R, d_x, d_y, tiles, m_end);
}

double bessel_i0(double x) {  // power series, converges fast for the beta used here
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) { term *= q / (static_cast<double>(k) * k); sum += term; if (term < 1e-18 * sum) break; }
    return sum;
}

int64_t gcd64(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }

}  // namespace

extern "C" {

int64_t fa_resample_linear_frames(int64_t frames, double in_rate, double out_rate) {
    if (frames < 0 || !(in_rate > 0) || !(out_rate > 0)) return 0;
    if (in_rate == out_rate) return frames;  // :414-416
    return static_cast<int64_t>(static_cast<double>(frames) / (in_rate / out_rate));  // :420
}

fa_status fa_resample_linear(fa_ctx *ctx, const float *planar, int32_t channels, int64_t frames, double in_rate, double out_rate,
                             float *out, int64_t out_capacity, int64_t *out_frames) {
    if (!ctx || !out_frames) return FA_INVALID_ARGUMENT;
    *out_frames = 0;
    if (channels < 1 || frames < 0 || !(in_rate > 0) || !(out_rate > 0)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "resample: bad arguments");
    const int64_t n_out = fa_resample_linear_frames(frames, in_rate, out_rate);
    if (n_out > out_capacity) return fa::set_error(ctx, FA_OUTPUT_TOO_SMALL, "resample: output buffer too small");
    *out_frames = n_out;
    if (frames == 0 || n_out == 0) return FA_SUCCESS;
    if (!planar || !out) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_in, d_mono, d_out;
    hipError_t e;
    do {
        if ((e = d_in.alloc(sizeof(float) * frames * channels)) != hipSuccess) break;
        if ((e = d_mono.alloc(sizeof(float) * frames)) != hipSuccess) break;
        if ((e = d_out.alloc(sizeof(float) * n_out)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_in.p, planar, sizeof(float) * frames * channels, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        hipLaunchKernelGGL(mixdown_kernel, dim3(static_cast<unsigned>((frames + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                           d_in.as<float>(), d_mono.as<float>(), channels, frames);
        const float *src = d_mono.as<float>();
        if (in_rate != out_rate) {
            hipLaunchKernelGGL(linear_kernel, dim3(static_cast<unsigned>((n_out + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                               d_mono.as<float>(), d_out.as<float>(), frames, n_out, in_rate / out_rate);
            src = d_out.as<float>();
        }
        if ((e = hipGetLastError()) != hipSuccess) break;
        if ((e = hipMemcpyAsync(out, src, sizeof(float) * n_out, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    return fa::hip_status(ctx, e, "fa_resample_linear");
}

int64_t fa_resample_poly_frames(int64_t frames, int32_t up, int32_t down) {
    if (frames < 0 || up < 1 || down < 1) return 0;
    const int64_t g = gcd64(up, down);
    const int64_t u = up / g, dn = down / g;
    return (frames * u + dn - 1) / dn;  // ceil(n * up / down)
}

fa_status fa_resample_poly_taps(int32_t up, int32_t down, float *taps, int64_t capacity, int64_t *n_taps, int64_t *pre_remove) {
    if (up < 1 || down < 1 || !n_taps || !pre_remove) return FA_INVALID_ARGUMENT;
    const int64_t g = gcd64(up, down);
    const int64_t u = up / g, dn = down / g, mx = u > dn ? u : dn;
    const int64_t half = 10 * mx, len = 2 * half + 1;
    const int64_t pre_pad = dn - half % dn;
    *n_taps = len + pre_pad;
    *pre_remove = (half + pre_pad) / dn;
    if (!taps) return FA_SUCCESS;
    if (capacity < *n_taps) return FA_OUTPUT_TOO_SMALL;
    try {
        // firwin(len, 1/mx, window=('kaiser', 5.0)) * up : windowed sinc, cut-off 1/mx of Nyquist, unit DC gain
        std::vector<double> h(len);
        const double fc = 1.0 / static_cast<double>(mx), alpha = 0.5 * (len - 1), beta = 5.0, i0b = bessel_i0(beta);
        double sum = 0.0;
        for (int64_t n = 0; n < len; ++n) {
            const double m = static_cast<double>(n) - alpha;
            const double a = M_PI * fc * m;
            const double sinc = m == 0.0 ? 1.0 : sin(a) / a;
            const double r = 2.0 * n / static_cast<double>(len - 1) - 1.0;
            const double w = bessel_i0(beta * sqrt(1.0 - r * r > 0 ? 1.0 - r * r : 0.0)) / i0b;
            h[n] = fc * sinc * w;
            sum += h[n];
        }
        for (int64_t n = 0; n < pre_pad; ++n) taps[n] = 0.0f;
        for (int64_t n = 0; n < len; ++n) taps[pre_pad + n] = static_cast<float>(h[n] / sum * static_cast<double>(u));
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    }
}

// Device-resident form: d_x (frames samples) -> d_y (fa_resample_poly_frames samples), enqueued on the context's stream.
fa_status fa_resample_poly_dev(fa_ctx *ctx, const float *d_x, int64_t frames, int32_t up, int32_t down, float *d_y, int64_t out_capacity, int64_t *out_frames) {
    if (!ctx || !out_frames) return FA_INVALID_ARGUMENT;
    *out_frames = 0;
    if (frames < 0 || up < 1 || down < 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "resample_poly: bad arguments");
    const int64_t g = gcd64(up, down);
    const int u = static_cast<int>(up / g), dn = static_cast<int>(down / g);
    const int64_t n_out = fa_resample_poly_frames(frames, up, down);
    if (n_out > out_capacity) return fa::set_error(ctx, FA_OUTPUT_TOO_SMALL, "resample_poly: output buffer too small");
    *out_frames = n_out;
    if (frames == 0) return FA_SUCCESS;
    if (!d_x || !d_y) return FA_INVALID_ARGUMENT;
    try {
        int64_t n_taps = 0, pre_remove = 0;
        FA_TRY(fa_resample_poly_taps(u, dn, nullptr, 0, &n_taps, &pre_remove));
        fa::DeviceGuard guard(ctx->device);
        // The Kaiser taps of a rate pair are computed and uploaded ONCE per context (a context-owned buffer, not the shared scratch):
        // a repeated call with the same (up, down) enqueues its kernel and returns without touching the host or synchronising.
        if (ctx->poly_up != u || ctx->poly_down != dn || !ctx->poly_taps) {
            std::vector<float> taps(n_taps);
            FA_TRY(fa_resample_poly_taps(u, dn, taps.data(), n_taps, &n_taps, &pre_remove));
            FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // an earlier call may still read the previous pair's taps
            if (ctx->poly_taps_bytes < sizeof(float) * n_taps) {
                if (ctx->poly_taps) { (void)hipFree(ctx->poly_taps); ctx->poly_taps = nullptr; ctx->poly_taps_bytes = 0; }
                FA_HIP_TRY(ctx, hipMalloc(&ctx->poly_taps, sizeof(float) * n_taps));
                ctx->poly_taps_bytes = sizeof(float) * n_taps;
            }
            ctx->poly_up = 0;
            FA_HIP_TRY(ctx, hipMemcpyAsync(ctx->poly_taps, taps.data(), sizeof(float) * n_taps, hipMemcpyHostToDevice, ctx->stream));
            FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // taps is a host temporary (first call of a rate pair only)
            ctx->poly_up = u; ctx->poly_down = dn;
            // per-phase tables of the pair for poly_rows_kernel (non-integer ratios), built and uploaded with the taps
            if (ctx->poly_rows && ctx->poly_rows_free) { ctx->poly_rows_free(ctx->poly_rows); ctx->poly_rows = nullptr; }
            {
                PolyRows *R = new PolyRows();
                std::vector<int> gtab;
                std::vector<float> tt;
                bool ok = poly_rows_build(*R, taps, u, dn, pre_remove, gtab, tt);
                if (ok) {
                    const size_t b0 = (sizeof(int) * gtab.size() + 255) & ~static_cast<size_t>(255), b1 = sizeof(float) * tt.size();
                    R->tt_offset = b0;
                    ok = hipMalloc(&R->d_tables, b0 + b1) == hipSuccess &&
                         hipMemcpyAsync(R->d_tables, gtab.data(), sizeof(int) * gtab.size(), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                         hipMemcpyAsync(static_cast<char *>(R->d_tables) + b0, tt.data(), b1, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                         hipStreamSynchronize(ctx->stream) == hipSuccess;
                    if (!ok) (void)hipGetLastError();
                }
                if (ok) { ctx->poly_rows = R; ctx->poly_rows_free = poly_rows_free; }
                else delete R;                                            // the LDS-staged kernel serves the pair
            }
        }
        float *d_h = static_cast<float *>(ctx->poly_taps);
        // the kernel-choice switches of the tests and probes, read ONCE per call, here (the forms fixed with a context's tables — FA_RESAMPLE_WIDE*, _ROWS_* —
        // are read when the tables are built)
        struct { bool simple, no_decim, no_decim_tiles, no_interp, no_rows; } const sw = {
            fa::sw_on(fa::Sw::RESAMPLE_SIMPLE), fa::sw_on(fa::Sw::RESAMPLE_NO_DECIM), fa::sw_on(fa::Sw::RESAMPLE_NO_DECIM_TILES),
            fa::sw_on(fa::Sw::RESAMPLE_NO_INTERP), fa::sw_on(fa::Sw::RESAMPLE_NO_ROWS)};
        const bool simple = sw.simple;
        auto edges = [&](const int64_t m_lo, const int64_t m_hi) {   // outputs [m_lo, m_hi) by the one-thread-per-output kernel (clamps at the signal's ends)
            if (m_hi <= m_lo) return;
            hipLaunchKernelGGL(poly_kernel, dim3(static_cast<unsigned>((m_hi - m_lo + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, frames, n_out,
                               n_taps, u, dn, pre_remove, m_lo, m_hi);
        };
        // integer decimation: register-tiled kernel on the outputs whose inputs all exist, poly_kernel on the two edges
        bool decim = false;
        if (!simple && u == 1 && ((dn >= 2 && dn <= 6) || dn == 12) && !sw.no_decim &&
            (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_y) & 7) == 0 && n_taps == 21 * dn + 1 && pre_remove == 11) {
            const int64_t m_begin = 10;                                                  // inputs start at (m - 10) dn >= 0; 10 = 10 (mod 4)
            const int64_t m_last = (frames - 1) / dn - 11;                               // (m + 11) dn <= frames - 1
            const int64_t avail = m_last >= m_begin ? std::min(m_last + 1, n_out) - m_begin : 0;   // outputs whose inputs all exist
            int64_t m_done = m_begin;                                                    // outputs [m_begin, m_done) are served by the tiled kernel
            if (!sw.no_decim_tiles) {
                // whole tiles of 256 R outputs through LDS (poly_decim_tile_kernel); what is left goes to the register-tiled kernel and the edges
                auto go = [&](auto dc) {
                    constexpr int DN = decltype(dc)::value;
                    typedef DecimTile<DN> D;
                    int64_t tiles = std::min<int64_t>(avail / D::TO, (int64_t{1} << 30) / D::TO);
                    // the 16-byte piece that holds a tile's last input may run up to 3 samples past it: a last tile whose piece would leave the signal is not a
                    // tile (its piece would be clamped and land shifted in LDS) — its outputs go to the register-tiled kernel and the edges (the same
                    // guard as the register-tiled kernel's below, mirrored in tests/cpu/resample_geom_emul.cpp)
                    while (tiles > 0 && ((m_begin + tiles * D::TO - 1) + 11) * DN + 3 > frames - 1) --tiles;
                    if (tiles <= 0) return;
                    const int per_cu = std::max(1, std::min(4, static_cast<int>(160 * 1024 / (2 * sizeof(float) * D::BUF))));
                    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(tiles, 256 * per_cu));
                    hipLaunchKernelGGL(poly_decim_tile_kernel<DN>, dim3(grid), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_begin, static_cast<int>(tiles), frames - 4);
                    m_done = m_begin + tiles * D::TO;
                };
                switch (dn) {
                    case 2: go(std::integral_constant<int, 2>{}); break;
                    case 3: go(std::integral_constant<int, 3>{}); break;
                    case 4: go(std::integral_constant<int, 4>{}); break;
                    case 5: go(std::integral_constant<int, 5>{}); break;
                    case 6: go(std::integral_constant<int, 6>{}); break;
                    default: go(std::integral_constant<int, 12>{}); break;          // 192 kHz: tiles only (no register-tiled instance: what is left goes to the edges' kernel)
                }
            }
            const int64_t m_rest = m_done;                                               // m_done - 10 is a multiple of 4 (tiles of 256 R outputs)
            const int64_t groups = dn <= 6 ? (avail - (m_rest - m_begin)) / kDecimR : 0;
            // the 16-byte loads of the last group may run up to 3 samples past its last input: keep them inside the signal
            int64_t gr = groups;
            while (gr > 0 && ((m_rest + gr * kDecimR - 1) + 11) * dn + 3 > frames - 1) --gr;
            if (gr > 0) {
                const unsigned grid = static_cast<unsigned>((gr + kThreads - 1) / kThreads);
                switch (dn) {
                    case 2: hipLaunchKernelGGL(poly_decim_kernel<2>, dim3(grid), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_rest, gr); break;
                    case 3: hipLaunchKernelGGL(poly_decim_kernel<3>, dim3(grid), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_rest, gr); break;
                    case 4: hipLaunchKernelGGL(poly_decim_kernel<4>, dim3(grid), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_rest, gr); break;
                    case 5: hipLaunchKernelGGL(poly_decim_kernel<5>, dim3(grid), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_rest, gr); break;
                    default: hipLaunchKernelGGL(poly_decim_kernel<6>, dim3(grid), dim3(kThreads), 0, ctx->stream, d_x, d_h, d_y, m_rest, gr); break;   // 96 kHz: 127 taps in eight registers, 169 inputs
                }
            }
            if (gr > 0 || m_done > m_begin) {
                edges(0, m_begin);
                edges(m_rest + gr * kDecimR, n_out);
                decim = true;
            }
        }
        // small interpolation factors: register-tiled kernel on the outputs whose inputs all exist, poly_kernel on the two ends
        if (!decim && !simple && u >= 2 && u <= 4 && !sw.no_interp) {
            int64_t lo = 0, hi = 0;
            if (u == 2 && dn == 1 && n_taps == 42) poly_interp_launch<2, 1, 42>(ctx, d_x, d_h, d_y, frames, n_out, pre_remove, lo, hi);
            else if (u == 2 && dn == 3 && n_taps == 64) poly_interp_launch<2, 3, 64>(ctx, d_x, d_h, d_y, frames, n_out, pre_remove, lo, hi);
            else if (u == 4 && dn == 3 && n_taps == 83) poly_interp_launch<4, 3, 83>(ctx, d_x, d_h, d_y, frames, n_out, pre_remove, lo, hi);
            else if (u == 4 && dn == 1 && n_taps == 82) poly_interp_launch<4, 1, 82>(ctx, d_x, d_h, d_y, frames, n_out, pre_remove, lo, hi);
            else if (u == 3 && dn == 1 && n_taps == 62) poly_interp_launch<3, 1, 62>(ctx, d_x, d_h, d_y, frames, n_out, pre_remove, lo, hi);
            else if (u == 3 && dn == 2 && n_taps == 63) poly_interp_launch<3, 2, 63>(ctx, d_x, d_h, d_y, frames, n_out, pre_remove, lo, hi);
            if (hi > lo) {
                edges(0, lo);
                edges(hi, n_out);
                decim = true;   // served: the kernels below have nothing left to do
            }
        }
        // non-integer ratios: row-tiled kernel on the tiles whose staged inputs all exist, poly_kernel on the two ends
        bool rows = false;
        if (!decim && !simple && ctx->poly_rows && !sw.no_rows) {
            const PolyRows &R = *static_cast<const PolyRows *>(ctx->poly_rows);
            const PolyRowsGeom &G = R.g;
            const int tile_rows = R.wide ? R.wide_rows : 64;
            int64_t tiles = fa::rows_tiles(G, frames, n_out, tile_rows);       // tiles whose staged inputs all exist (resample_geom.h)
            const int64_t per_tile = static_cast<int64_t>(tile_rows) * G.up;
            if (tiles > 0 && (tiles + 8) * G.groups < (1LL << 31)) {
                const int64_t m_stop = std::min(n_out, G.m_begin + tiles * per_tile);
                if (R.wide) poly_rows_wide_launch(ctx, R, d_x, d_y, tiles, m_stop, frames);
                else switch (R.nv) {
#define FA_ROWS_CASE(V) case V: poly_rows_launch<V>(ctx, R, d_x, d_y, tiles, m_stop, frames); break;
                    FA_ROWS_CASE(4) FA_ROWS_CASE(6) FA_ROWS_CASE(8) FA_ROWS_CASE(10) FA_ROWS_CASE(12) FA_ROWS_CASE(14) FA_ROWS_CASE(16) FA_ROWS_CASE(32)
#undef FA_ROWS_CASE
                    default: tiles = 0; break;
                }
                if (tiles > 0) {
                    edges(0, G.m_begin);
                    edges(m_stop, n_out);
                    rows = true;
                }
            }
        }
        const int64_t span = (static_cast<int64_t>(kPolyTile) * dn + u - 1) / u + (n_taps + u - 1) / u + 4;
        const size_t lds = sizeof(float) * (static_cast<size_t>((n_taps + 3) & ~static_cast<int64_t>(3)) + static_cast<size_t>(span));
        if (decim || rows) {
        } else if (lds <= 150 * 1024 && !simple) {
            if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(poly_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            const int64_t tiles = (n_out + kPolyTile - 1) / kPolyTile;
            const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
            const int grid = static_cast<int>(tiles < 256 * per_cu ? tiles : 256 * per_cu);
            hipLaunchKernelGGL(poly_lds_kernel, dim3(grid), dim3(kThreads), lds, ctx->stream, d_x, d_h, d_y, frames, n_out, static_cast<int>(n_taps), u, dn, pre_remove,
                               static_cast<int>(span));
        } else {   // very long filters (extreme rate ratios): one thread per output straight from global memory
            edges(0, n_out);
        }
        FA_HIP_TRY(ctx, hipGetLastError());
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fa_status fa_resample_poly(fa_ctx *ctx, const float *x, int64_t frames, int32_t up, int32_t down, float *out, int64_t out_capacity,
                           int64_t *out_frames) {
    if (!ctx || !out_frames) return FA_INVALID_ARGUMENT;
    *out_frames = 0;
    if (frames < 0 || up < 1 || down < 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "resample_poly: bad arguments");
    const int64_t n_out = fa_resample_poly_frames(frames, up, down);
    if (n_out > out_capacity) return fa::set_error(ctx, FA_OUTPUT_TOO_SMALL, "resample_poly: output buffer too small");
    *out_frames = n_out;
    if (frames == 0) return FA_SUCCESS;
    if (!x || !out) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_x, d_y;
    if (d_x.alloc(sizeof(float) * frames) != hipSuccess || d_y.alloc(sizeof(float) * n_out) != hipSuccess) {
        (void)hipGetLastError();
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "resample_poly: device allocation failed");
    }
    FA_HIP_TRY(ctx, hipMemcpyAsync(d_x.p, x, sizeof(float) * frames, hipMemcpyHostToDevice, ctx->stream));
    int64_t got = 0;
    FA_TRY(fa_resample_poly_dev(ctx, d_x.as<float>(), frames, up, down, d_y.as<float>(), n_out, &got));
    FA_HIP_TRY(ctx, hipMemcpyAsync(out, d_y.p, sizeof(float) * n_out, hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FA_SUCCESS;
}

}  // extern "C"
