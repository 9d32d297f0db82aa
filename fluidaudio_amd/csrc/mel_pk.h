// mel_pk.h — frame-pair packed arithmetic of the STFT->mel kernel (device only, gfx950).
//
// Same dataflow as mel_core.h (two radix-16 passes in registers, one LDS transpose, DPP partner exchange, even/odd
// recombination), but every lane carries TWO consecutive frames in the halves of 64-bit register pairs and every
// add / multiply / fma is a packed-fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two fp32 results per lane at
// the issue cost of one) — the kernel is VALU-issue bound, so halving the instruction count per frame is the
// lever.  Packing by FRAME (not by re/im) keeps every value in the same half from load to store: no half swaps, and constants
// (window, twiddles) are shared by both halves.  A lane constant pair (c0, c1) occupies ONE register pair and is broadcast to
// both halves by the op_sel / op_sel_hi source selectors of the packed instructions — the compiler does not emit those for a
// scalar * vector product (it materialises a (c, c) pair per constant: 2 x 78 registers), hence the inline-asm wrappers.
// The arithmetic (everything except the DPP exchange) also compiles for the host with clang (ext_vector_type), where the
// op_sel wrappers fall back to plain vector code: tests/cpu/mel_pk_emul.cpp replays the 16 lanes of a frame group on the CPU.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FA_PK __device__ __forceinline__
#else
#define FA_PK inline
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
#endif

#include "mel_core.h"

namespace fa {
namespace melpk {

using melcore::kEStride;

typedef float f2 __attribute__((ext_vector_type(2)));

#if !defined(__HIP_DEVICE_COMPILE__)
// host restatement of the op_sel-broadcast wrappers below (the host pass of hipcc never runs them; the CPU replay does)
FA_PK f2 mul_lo(const f2 a, const f2 w) { return a * w.x; }
FA_PK f2 mul_hi(const f2 a, const f2 w) { return a * w.y; }
FA_PK f2 fma_lo(const f2 a, const f2 w, const f2 c) { return a * w.x + c; }
FA_PK f2 fma_hi(const f2 a, const f2 w, const f2 c) { return a * w.y + c; }
FA_PK f2 fms_lo(const f2 a, const f2 w, const f2 c) { return a * w.x - c; }
FA_PK f2 fms_hi(const f2 a, const f2 w, const f2 c) { return a * w.y - c; }
#else
// d = a * w.lo / a * w.hi (both halves of a times the SAME scalar half of w)
__device__ __forceinline__ f2 mul_lo(const f2 a, const f2 w) { f2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(w)); return d; }
__device__ __forceinline__ f2 mul_hi(const f2 a, const f2 w) { f2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(w)); return d; }
// d = a * w.lo + c, a * w.hi + c, a * w.lo - c, a * w.hi - c
__device__ __forceinline__ f2 fma_lo(const f2 a, const f2 w, const f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(c)); return d; }
__device__ __forceinline__ f2 fma_hi(const f2 a, const f2 w, const f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(w), "v"(c)); return d; }
__device__ __forceinline__ f2 fms_lo(const f2 a, const f2 w, const f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(c)); return d; }
__device__ __forceinline__ f2 fms_hi(const f2 a, const f2 w, const f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(c)); return d; }
#endif

struct LanePk {
    f2 re[16];
    f2 im[16];
};

// per-lane twiddle constants, two per register pair.  The 32 window values of a lane live in LDS (window_table below): they are
// used once per pass, right after the sample reads, and 32 more resident registers push the kernel into spills.
struct LaneConstPk {
    f2 t1[15];  // exp(-2 pi i lane k1 / 256) as (re, im), k1 = 1..15 at index k1 - 1
    f2 t2[8];   // exp(-2 pi i (lane + 16 j) / 512) as (re, im)
};

constexpr int kWindowTableFloats = 8 * 16 * 4;
// LDS window table: float4 entry (q, lane) = windowz[32 (2 q) + 2 lane + {0, 1}], windowz[32 (2 q + 1) + 2 lane + {0, 1}];
// a 16-lane group reads 256 contiguous bytes per q (conflict-free), all groups of a wavefront the same addresses (broadcast).
FA_PK void window_table_fill(const int tid, const int threads, const float *windowz, float *tab) {
    for (int i = tid; i < kWindowTableFloats; i += threads) {
        const int c = i & 3, lane = (i >> 2) & 15, q = i >> 6;
        tab[i] = windowz[32 * (2 * q + (c >> 1)) + 2 * lane + (c & 1)];
    }
}

FA_PK void lane_const_init(const int lane, const float2 *tw256, const float2 *tw512, LaneConstPk &k) {
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) { const float2 t = tw256[(lane * k1) & 255]; k.t1[k1 - 1] = f2{t.x, t.y}; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float2 t = tw512[lane + 16 * j]; k.t2[j] = f2{t.x, t.y}; }
}

FA_PK void fft4(f2 &r0, f2 &i0, f2 &r1, f2 &i1, f2 &r2, f2 &i2, f2 &r3, f2 &i3) {
    const f2 ar = r0 + r2, ai = i0 + i2;
    const f2 br = r0 - r2, bi = i0 - i2;
    const f2 cr = r1 + r3, ci = i1 + i3;
    const f2 dr = i1 - i3, di = r3 - r1;  // (x1 - x3) * (-i)
    r0 = ar + cr; i0 = ai + ci;
    r2 = ar - cr; i2 = ai - ci;
    r1 = br + dr; i1 = bi + di;
    r3 = br - dr; i3 = bi - di;
}

// (r + i i) * (w.lo + i w.hi), w a lane constant pair
FA_PK void cmul_w(f2 &r, f2 &i, const f2 w) {
    const f2 tr = fms_lo(r, w, mul_hi(i, w));  // r wr - i wi
    const f2 ti = fma_hi(r, w, mul_lo(i, w));  // r wi + i wr
    r = tr; i = ti;
}

FA_PK void cmul_c(f2 &r, f2 &i, const float wr, const float wi) {  // literal twiddles of the 16-point DFT
    const f2 tr = r * wr - i * wi;
    const f2 ti = r * wi + i * wr;
    r = tr; i = ti;
}

// in-register 16-point forward DFT, natural order in and out (same factorisation as melcore::fft16)
FA_PK void fft16(LanePk &v) {
    constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f, R = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2)
        fft4(v.re[n2], v.im[n2], v.re[4 + n2], v.im[4 + n2], v.re[8 + n2], v.im[8 + n2], v.re[12 + n2], v.im[12 + n2]);
    cmul_c(v.re[5], v.im[5], C, -S);
    cmul_c(v.re[6], v.im[6], R, -R);
    cmul_c(v.re[7], v.im[7], S, -C);
    cmul_c(v.re[9], v.im[9], R, -R);
    { const f2 t = v.re[10]; v.re[10] = v.im[10]; v.im[10] = -t; }
    cmul_c(v.re[11], v.im[11], -R, -R);
    cmul_c(v.re[13], v.im[13], S, -C);
    cmul_c(v.re[14], v.im[14], -R, -R);
    cmul_c(v.re[15], v.im[15], -C, S);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
        fft4(v.re[4 * k1], v.im[4 * k1], v.re[4 * k1 + 1], v.im[4 * k1 + 1], v.re[4 * k1 + 2], v.im[4 * k1 + 2], v.re[4 * k1 + 3],
             v.im[4 * k1 + 3]);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            f2 t = v.re[4 * a + b]; v.re[4 * a + b] = v.re[4 * b + a]; v.re[4 * b + a] = t;
            t = v.im[4 * a + b]; v.im[4 * a + b] = v.im[4 * b + a]; v.im[4 * b + a] = t;
        }
}

// window, first radix-16 pass, inter-pass twiddle, transpose through the group's LDS region (real parts, then imaginary parts:
// the region holds 16 x 17 frame pairs = the footprint of ONE frame of the scalar kernel), second radix-16 pass.
// On entry v holds the raw samples 32 n1 + 2 lane (re) and + 1 (im) of both frames and w4[q] the lane's window_table row;
// on exit v.re/im[k2] = Z[lane + 16 k2].  EZ: the window is zero on the first and the last 32 positions of the frame (n1 = 0
// and 15; true for a 400-sample window centred in 512): those blocks are literal zeros, never loaded or multiplied, and the
// additions they feed in the first 16-point pass fold away.
template <bool EZ>
FA_PK void fft256_head(LanePk &v, const float4 (&w4)[8], const LaneConstPk &k) {   // window, first radix-16 pass, inter-pass twiddle
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f2 wa = f2{w4[q].x, w4[q].y}, wb = f2{w4[q].z, w4[q].w};
        if (!(EZ && q == 0)) { v.re[2 * q] = mul_lo(v.re[2 * q], wa); v.im[2 * q] = mul_hi(v.im[2 * q], wa); }
        if (!(EZ && q == 7)) { v.re[2 * q + 1] = mul_lo(v.re[2 * q + 1], wb); v.im[2 * q + 1] = mul_hi(v.im[2 * q + 1], wb); }
    }
    fft16(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) cmul_w(v.re[k1], v.im[k1], k.t1[k1 - 1]);
}
FA_PK void transpose_put(const int lane, const f2 (&c)[16], f2 *region) {
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) region[k1 * kEStride + lane] = c[k1];
}
FA_PK void transpose_get(const int lane, f2 (&c)[16], const f2 *region) {
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) c[n2] = region[lane * kEStride + n2];
}
template <bool EZ>
FA_PK void fft256(const int lane, LanePk &v, const float4 (&w4)[8], const LaneConstPk &k, f2 *region) {
    fft256_head<EZ>(v, w4, k);
    // LDS operations of one wavefront complete in program order: write re, read re, write im (over re), read im
    transpose_put(lane, v.re, region);
    transpose_get(lane, v.re, region);
    transpose_put(lane, v.im, region);
    transpose_get(lane, v.im, region);
    fft16(v);
}

#if defined(__HIPCC__)
// lane l <- lane (16 - l) & 15 inside every row of 16 lanes (row_mirror, then row_ror:1), both halves
__device__ __forceinline__ f2 partner(const f2 x) {
    int a = __float_as_int(x.x), b = __float_as_int(x.y);
    a = __builtin_amdgcn_update_dpp(0, a, 0x140, 0xf, 0xf, true);
    b = __builtin_amdgcn_update_dpp(0, b, 0x140, 0xf, 0xf, true);
    a = __builtin_amdgcn_update_dpp(0, a, 0x121, 0xf, 0xf, true);
    b = __builtin_amdgcn_update_dpp(0, b, 0x121, 0xf, 0xf, true);
    return f2{__int_as_float(a), __int_as_float(b)};
}
// the same exchange with lane 0 of every row keeping `own` (its partner is one of its own registers): row_mirror, then
// row_shr:1 — a shift has no source for lane 0, which therefore keeps the `old` operand.  Saves the select per value.
__device__ __forceinline__ f2 partner_or(const f2 x, const f2 own) {
    int a = __float_as_int(x.x), b = __float_as_int(x.y);
    a = __builtin_amdgcn_update_dpp(0, a, 0x140, 0xf, 0xf, true);
    b = __builtin_amdgcn_update_dpp(0, b, 0x140, 0xf, 0xf, true);
    a = __builtin_amdgcn_update_dpp(__float_as_int(own.x), a, 0x111, 0xf, 0xf, false);
    b = __builtin_amdgcn_update_dpp(__float_as_int(own.y), b, 0x111, 0xf, 0xf, false);
    return f2{__int_as_float(a), __int_as_float(b)};
}
#endif

// 4 |X[k]|^2 and 4 |X[256 - k]|^2 from A = Z[k], B = conj(Z[256 - k]) (bi_src = Im Z[256 - k]), w = exp(-2 pi i k / 512):
// 2 X[k] = (A + B) - i w (A - B), 2 conj(X[256 - k]) = (A + B) + i w (A - B).  The factor 4 (exact) is folded into the filterbank.
FA_PK void pair_power4(const f2 ar, const f2 ai, const f2 br, const f2 bi_src, const f2 w, f2 &p_lo, f2 &p_hi) {
    const f2 sr = ar + br, si = ai - bi_src;
    const f2 dr = ar - br, di = ai + bi_src;
    const f2 a1 = fma_lo(di, w, mul_hi(dr, w));   // wr di + wi dr
    const f2 b1 = fms_lo(dr, w, mul_hi(di, w));   // wr dr - wi di
    const f2 xr = sr + a1, xi = si - b1;
    const f2 yr = sr - a1, yi = si + b1;
    p_lo = xr * xr + xi * xi;
    p_hi = yr * yr + yi * yi;
}

}  // namespace melpk
}  // namespace fa
