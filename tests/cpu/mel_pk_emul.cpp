// Host replay of the frame-pair packed dataflow (fluidaudio_amd/csrc/mel_pk.h, the same source the GPU compiles; the op_sel
// wrappers fall back to plain vector code and the DPP partner exchange is the index map of melcore::partner_of).
// One C function: two 512-sample frames in, the 257 power bins of each out (4 |X|^2 as the kernel stores them, times 1/4).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../fluidaudio_amd/csrc/mel_pk.h"
using namespace fa::melpk;

template <bool EZ>
static void run(const float *frame_a, const float *frame_b, const float *windowz, float *power_a, float *power_b) {
    std::vector<float2> t256(256), t512(129);
    for (int k = 0; k < 256; ++k) { const double a = -2.0 * M_PI * k / 256.0; t256[k] = float2{(float)cos(a), (float)sin(a)}; }
    for (int k = 0; k < 129; ++k) { const double a = -2.0 * M_PI * k / 512.0; t512[k] = float2{(float)cos(a), (float)sin(a)}; }
    std::vector<float> wtab(kWindowTableFloats);
    window_table_fill(0, 1, windowz, wtab.data());
    std::vector<f2> region(16 * fa::melcore::kEStride + 1);
    LanePk v[16];
    LaneConstPk kc[16];
    float4 w4[16][8];
    for (int lane = 0; lane < 16; ++lane) {
        lane_const_init(lane, t256.data(), t512.data(), kc[lane]);
        for (int q = 0; q < 8; ++q) std::memcpy(&w4[lane][q], wtab.data() + (q * 16 + lane) * 4, sizeof(float4));
        for (int n1 = 0; n1 < 16; ++n1) {
            v[lane].re[n1] = f2{frame_a[32 * n1 + 2 * lane], frame_b[32 * n1 + 2 * lane]};
            v[lane].im[n1] = f2{frame_a[32 * n1 + 2 * lane + 1], frame_b[32 * n1 + 2 * lane + 1]};
        }
        if (EZ) { v[lane].re[0] = v[lane].im[0] = v[lane].re[15] = v[lane].im[15] = f2{0.0f, 0.0f}; }
        fft256_head<EZ>(v[lane], w4[lane], kc[lane]);
    }
    // the four LDS phases of fft256, every lane of the group per phase
    for (int lane = 0; lane < 16; ++lane) transpose_put(lane, v[lane].re, region.data());
    for (int lane = 0; lane < 16; ++lane) transpose_get(lane, v[lane].re, region.data());
    for (int lane = 0; lane < 16; ++lane) transpose_put(lane, v[lane].im, region.data());
    for (int lane = 0; lane < 16; ++lane) transpose_get(lane, v[lane].im, region.data());
    for (int lane = 0; lane < 16; ++lane) fft16(v[lane]);
    std::vector<f2> P(257);
    for (int lane = 0; lane < 16; ++lane) {
        for (int j = 0; j < 8; ++j) {
            int pl, pr;
            fa::melcore::partner_of(lane, j, pl, pr);
            f2 lo, hi;
            pair_power4(v[lane].re[j], v[lane].im[j], v[pl].re[pr], v[pl].im[pr], kc[lane].t2[j], lo, hi);
            P[lane + 16 * j] = lo;
            P[256 - (lane + 16 * j)] = hi;
        }
    }
    P[128] = 4.0f * (v[0].re[8] * v[0].re[8] + v[0].im[8] * v[0].im[8]);
    for (int k = 0; k < 257; ++k) { power_a[k] = 0.25f * P[k].x; power_b[k] = 0.25f * P[k].y; }
}

extern "C" void mel_pk_emul_power(const float *frame_a, const float *frame_b, const float *windowz, int edge_zero, float *power_a, float *power_b) {
    if (edge_zero) run<true>(frame_a, frame_b, windowz, power_a, power_b);
    else run<false>(frame_a, frame_b, windowz, power_a, power_b);
}
