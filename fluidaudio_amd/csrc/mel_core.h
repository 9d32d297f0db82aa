// mel_core.h — per-lane arithmetic of the STFT->mel kernel (fluidaudio_amd/csrc/mel.hip).
//
// The 512-point real DFT of one frame (AudioMelSpectrogram.swift:459-481 computes it with
// vDSP_DFT_zop) is evaluated by a 16-lane group as a 256-point complex FFT of
// z[n] = x[2n] + i*x[2n+1] (two radix-16 passes held in registers, one transpose through
// the group's LDS region) followed by the even/odd recombination that yields the 257
// power bins.  The recombination pairs Z[k] with Z[256-k]; with k = lane + 16*j the partner
// lives in lane (16 - lane) & 15, so that exchange is a lane permutation (DPP row_mirror +
// row_ror:1 on the device) and never touches LDS.  Window and twiddle factors a lane needs
// depend only on (lane & 15): they are loaded once into registers (LaneConst).
// Every function here is written against (lane, LDS region, registers) so the same code
// runs inside the kernel and inside tests/cpu/mel_core_emul.cpp, which replays the 16 lanes
// sequentially on the host to check the index algebra without a GPU.
#pragma once

#if defined(__HIPCC__)
#define FA_HD __host__ __device__ __forceinline__
#else
#define FA_HD inline
#endif

namespace fa {
namespace melcore {

constexpr int kNfft = 512;          // real DFT length (fixed in this kernel generation)
constexpr int kHalf = 256;          // complex FFT length
constexpr int kBins = 257;          // power bins 0..256
constexpr int kGroup = 16;          // lanes per frame
constexpr int kEStride = 17;        // complex elements per row of the transpose buffer (bank spread)
constexpr int kRegionFloats = 578;  // floats per frame region: >=2*16*17, ==2 (mod 64) => see DESIGN.md

struct Lane {
    float re[16];
    float im[16];
};

// Constant tables shared by every lane (device: global memory, L1/L2 resident; ~5 KB).
struct Tables {
    const float *windowz;  // [512]   analysis window zero-extended to n_fft at its frame offset
    const float *tw256;    // [256*2] exp(-2*pi*i*k/256) as (re, im) pairs
    const float *tw512;    // [129*2] exp(-2*pi*i*k/512) as (re, im) pairs
};

// ---- in-register 16-point forward DFT (natural order in -> natural order out) -------------
FA_HD void fft4(float &r0, float &i0, float &r1, float &i1, float &r2, float &i2, float &r3, float &i3) {
    const float ar = r0 + r2, ai = i0 + i2;
    const float br = r0 - r2, bi = i0 - i2;
    const float cr = r1 + r3, ci = i1 + i3;
    const float dr = i1 - i3, di = r3 - r1;  // (x1 - x3) * (-i)
    r0 = ar + cr; i0 = ai + ci;
    r2 = ar - cr; i2 = ai - ci;
    r1 = br + dr; i1 = bi + di;
    r3 = br - dr; i3 = bi - di;
}

FA_HD void cmul(float &r, float &i, const float wr, const float wi) {
    const float tr = r * wr - i * wi;
    const float ti = r * wi + i * wr;
    r = tr; i = ti;
}

FA_HD void fft16(Lane &v) {
    constexpr float C = 0.92387953251128674f;  // cos(pi/8)
    constexpr float S = 0.38268343236508977f;  // sin(pi/8)
    constexpr float R = 0.70710678118654752f;  // sqrt(1/2)
    // step A: for each n2, 4-point DFT over n1 of x[4*n1 + n2]; result A[n2][k1] stored at index 4*k1 + n2
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int n2 = 0; n2 < 4; ++n2)
        fft4(v.re[n2], v.im[n2], v.re[4 + n2], v.im[4 + n2], v.re[8 + n2], v.im[8 + n2], v.re[12 + n2], v.im[12 + n2]);
    // twiddle W16^(n2*k1) on element index 4*k1 + n2
    cmul(v.re[5], v.im[5], C, -S);     // n2=1,k1=1 : W^1
    cmul(v.re[6], v.im[6], R, -R);     // n2=2,k1=1 : W^2
    cmul(v.re[7], v.im[7], S, -C);     // n2=3,k1=1 : W^3
    cmul(v.re[9], v.im[9], R, -R);     // n2=1,k1=2 : W^2
    { const float t = v.re[10]; v.re[10] = v.im[10]; v.im[10] = -t; }  // n2=2,k1=2 : W^4 = -i
    cmul(v.re[11], v.im[11], -R, -R);  // n2=3,k1=2 : W^6
    cmul(v.re[13], v.im[13], S, -C);   // n2=1,k1=3 : W^3
    cmul(v.re[14], v.im[14], -R, -R);  // n2=2,k1=3 : W^6
    cmul(v.re[15], v.im[15], -C, S);   // n2=3,k1=3 : W^9
    // step B: for each k1, 4-point DFT over n2 (indices 4*k1 + n2); output X[k1 + 4*k2] lands at 4*k1 + k2
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k1 = 0; k1 < 4; ++k1)
        fft4(v.re[4 * k1], v.im[4 * k1], v.re[4 * k1 + 1], v.im[4 * k1 + 1], v.re[4 * k1 + 2], v.im[4 * k1 + 2],
             v.re[4 * k1 + 3], v.im[4 * k1 + 3]);
    // un-permute: element 4*k1 + k2 holds X[k1 + 4*k2]  -> transpose the 4x4 index grid
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int a = 0; a < 4; ++a)
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int b = a + 1; b < 4; ++b) {
            float t = v.re[4 * a + b]; v.re[4 * a + b] = v.re[4 * b + a]; v.re[4 * b + a] = t;
            t = v.im[4 * a + b]; v.im[4 * a + b] = v.im[4 * b + a]; v.im[4 * b + a] = t;
        }
}

// ---- phase A: window, first radix-16 pass, inter-pass twiddle, scatter into transpose buffer
// fs: the frame's 512-sample span in (LDS) sample storage, fs[j] = preemphasised sample at
//     frame position j (positions outside the analysis window are multiplied by hw == 0).
FA_HD void phase_a(const int lane, const float *fs, const Tables &c, float *region) {
    Lane v;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int n1 = 0; n1 < 16; ++n1) {
        const int j = 32 * n1 + 2 * lane;
        v.re[n1] = fs[j] * c.windowz[j];
        v.im[n1] = fs[j + 1] * c.windowz[j + 1];
    }
    fft16(v);  // v[k1] = sum_n1 z[16*n1 + lane] W16^(n1*k1)
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k1 = 0; k1 < 16; ++k1) {
        const int tk = (lane * k1) & (kHalf - 1);  // W256^(lane*k1)
        cmul(v.re[k1], v.im[k1], c.tw256[2 * tk], c.tw256[2 * tk + 1]);
        region[2 * (k1 * kEStride + lane)] = v.re[k1];
        region[2 * (k1 * kEStride + lane) + 1] = v.im[k1];
    }
}

// ---- phase B1: gather row `lane` (= k1) of the transpose buffer, second radix-16 pass.
FA_HD void phase_b1(const int lane, const float *region, Lane &v) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int n2 = 0; n2 < 16; ++n2) {
        v.re[n2] = region[2 * (lane * kEStride + n2)];
        v.im[n2] = region[2 * (lane * kEStride + n2) + 1];
    }
    fft16(v);  // v[k2] = Z[lane + 16*k2]
}

// ---- phase B2: store Z in natural order (complex index k at floats 2k, 2k+1).
FA_HD void phase_b2(const int lane, const Lane &v, float *region) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k2 = 0; k2 < 16; ++k2) {
        region[2 * (lane + 16 * k2)] = v.re[k2];
        region[2 * (lane + 16 * k2) + 1] = v.im[k2];
    }
}

struct Power {
    float lo[8];  // P[k],      k = lane + 16*j
    float hi[8];  // P[256 - k]
    float mid;    // P[128] (lane 0 only)
};

FA_HD void pair_power(const float ar, const float ai, const float br, const float bi_conj_src,
                      const float wr, const float wi, float &p_lo, float &p_hi) {
    // A = Z[k], B = conj(Z[256-k]) = (br, -bi_conj_src)
    const float bi = -bi_conj_src;
    const float sr = 0.5f * (ar + br), si = 0.5f * (ai + bi);
    const float dr = ar - br, di = ai - bi;
    const float vr = -0.5f * (wr * di + wi * dr);  // (i/2) * w * (A - B)
    const float vi = 0.5f * (wr * dr - wi * di);
    const float xr = sr - vr, xi = si - vi;        // X[k]
    const float yr = sr + vr, yi = si + vi;        // conj(X[256-k])
    p_lo = xr * xr + xi * xi;
    p_hi = yr * yr + yi * yi;
}

// ---- phase C1: even/odd recombination -> power bins, kept in registers.
FA_HD void phase_c1(const int lane, const float *region, const Tables &c, Power &p) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j) {
        const int k = lane + 16 * j;
        const int kp = (kHalf - k) & (kHalf - 1);
        pair_power(region[2 * k], region[2 * k + 1], region[2 * kp], region[2 * kp + 1], c.tw512[2 * k],
                   c.tw512[2 * k + 1], p.lo[j], p.hi[j]);
    }
    p.mid = 0.0f;
    if (lane == 0) {
        float dummy;
        // k = 128: w = exp(-i*pi/2) = (0, -1)
        pair_power(region[2 * 128], region[2 * 128 + 1], region[2 * 128], region[2 * 128 + 1], 0.0f, -1.0f, p.mid, dummy);
    }
}

// ---- phase C2: write the 257 power bins over the region (float index k).
FA_HD void phase_c2(const int lane, const Power &p, float *region) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j) {
        const int k = lane + 16 * j;
        region[k] = p.lo[j];
        region[kHalf - k] = p.hi[j];
    }
    if (lane == 0) region[128] = p.mid;
}

// =============================================================================== v2 dataflow
// Per-lane constants (lane = index inside the 16-lane frame group).
struct LaneConst {
    float wz[32];          // windowz[32*n1 + 2*lane], windowz[32*n1 + 2*lane + 1]  (n1 = 0..15)
    float t1r[15], t1i[15];  // exp(-2*pi*i*lane*k1/256), k1 = 1..15 at index k1 - 1
    float t2r[8], t2i[8];    // exp(-2*pi*i*(lane + 16*j)/512), j = 0..7
};

FA_HD void lane_const_init(const int lane, const Tables &c, LaneConst &k) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int n1 = 0; n1 < 16; ++n1) {
        k.wz[2 * n1] = c.windowz[32 * n1 + 2 * lane];
        k.wz[2 * n1 + 1] = c.windowz[32 * n1 + 2 * lane + 1];
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k1 = 1; k1 < 16; ++k1) {
        const int tk = (lane * k1) & (kHalf - 1);
        k.t1r[k1 - 1] = c.tw256[2 * tk];
        k.t1i[k1 - 1] = c.tw256[2 * tk + 1];
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j) {
        k.t2r[j] = c.tw512[2 * (lane + 16 * j)];
        k.t2i[j] = c.tw512[2 * (lane + 16 * j) + 1];
    }
}

// ---- phase A: window (registers), first radix-16 pass, inter-pass twiddle, scatter into the transpose buffer.
// v.re[n1] / v.im[n1] hold the frame samples 32*n1 + 2*lane, + 1 on entry.
FA_HD void phase_a2(const int lane, Lane &v, const LaneConst &k, float *region) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int n1 = 0; n1 < 16; ++n1) { v.re[n1] *= k.wz[2 * n1]; v.im[n1] *= k.wz[2 * n1 + 1]; }
    fft16(v);  // v[k1] = sum_n1 z[16*n1 + lane] W16^(n1*k1)
    region[2 * lane] = v.re[0];
    region[2 * lane + 1] = v.im[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k1 = 1; k1 < 16; ++k1) {
        cmul(v.re[k1], v.im[k1], k.t1r[k1 - 1], k.t1i[k1 - 1]);
        region[2 * (k1 * kEStride + lane)] = v.re[k1];
        region[2 * (k1 * kEStride + lane) + 1] = v.im[k1];
    }
}

// ---- phase C (v2): v = Z[lane + 16*k2] of this lane (after phase_b1); (qr, qi)[j] = Z[256 - (lane + 16*j)],
// fetched from the partner lane by the caller (device: DPP, host: array).  Power bins stay in registers.
FA_HD void phase_c1v2(const int lane, const Lane &v, const float (&qr)[8], const float (&qi)[8], const LaneConst &k, Power &p) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j)
        pair_power(v.re[j], v.im[j], qr[j], qi[j], k.t2r[j], k.t2i[j], p.lo[j], p.hi[j]);
    p.mid = 0.0f;
    if (lane == 0) {
        float dummy;
        pair_power(v.re[8], v.im[8], v.re[8], v.im[8], 0.0f, -1.0f, p.mid, dummy);  // k = 128: w = exp(-i*pi/2)
    }
}

// Which register of which lane holds Z[256 - (lane + 16*j)]: lane (16 - lane) & 15, register 15 - j; lane 0 pairs
// with itself, register (16 - j) & 15.  (Used by the host replay; the device hard-wires the same mapping in DPP.)
FA_HD void partner_of(const int lane, const int j, int &plane, int &preg) {
    plane = (16 - lane) & 15;
    preg = lane == 0 ? ((16 - j) & 15) : 15 - j;
}

}  // namespace melcore
}  // namespace fa
