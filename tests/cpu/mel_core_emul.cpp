// Host replay of the mel kernel's 16-lane dataflow (fluidaudio_amd/csrc/mel_core.h).
// Exposes one C function so tests/test_mel_core_emul.py can compare it with a float64 DFT.
#include <cmath>
#include <cstring>
#include <vector>
#include "../../fluidaudio_amd/csrc/mel_core.h"
using namespace fa::melcore;

extern "C" void mel_core_emul_power(const float* frame512 /*samples, un-windowed*/, const float* windowz512,
                                    float* power257) {
    std::vector<float> region(kRegionFloats, 0.f), t256(512), t512(258);
    for (int k = 0; k < 256; ++k) { const double a = -2.0 * M_PI * k / 256.0; t256[2 * k] = (float)cos(a); t256[2 * k + 1] = (float)sin(a); }
    for (int k = 0; k < 129; ++k) { const double a = -2.0 * M_PI * k / 512.0; t512[2 * k] = (float)cos(a); t512[2 * k + 1] = (float)sin(a); }
    Tables c{windowz512, t256.data(), t512.data()};
    for (int lane = 0; lane < 16; ++lane) phase_a(lane, frame512, c, region.data());
    Lane v[16];
    for (int lane = 0; lane < 16; ++lane) phase_b1(lane, region.data(), v[lane]);
    for (int lane = 0; lane < 16; ++lane) phase_b2(lane, v[lane], region.data());
    Power p[16];
    for (int lane = 0; lane < 16; ++lane) phase_c1(lane, region.data(), c, p[lane]);
    for (int lane = 0; lane < 16; ++lane) phase_c2(lane, p[lane], region.data());
    std::memcpy(power257, region.data(), sizeof(float) * kBins);
}

// v2 dataflow: register-resident lane constants, partner exchange by lane permutation (no LDS round trip).
extern "C" void mel_core_emul_power_v2(const float* frame512, const float* windowz512, float* power257) {
    std::vector<float> region(kRegionFloats, 0.f), t256(512), t512(258);
    for (int k = 0; k < 256; ++k) { const double a = -2.0 * M_PI * k / 256.0; t256[2 * k] = (float)cos(a); t256[2 * k + 1] = (float)sin(a); }
    for (int k = 0; k < 129; ++k) { const double a = -2.0 * M_PI * k / 512.0; t512[2 * k] = (float)cos(a); t512[2 * k + 1] = (float)sin(a); }
    Tables c{windowz512, t256.data(), t512.data()};
    LaneConst lc[16];
    Lane v[16];
    for (int lane = 0; lane < 16; ++lane) {
        lane_const_init(lane, c, lc[lane]);
        for (int n1 = 0; n1 < 16; ++n1) { v[lane].re[n1] = frame512[32 * n1 + 2 * lane]; v[lane].im[n1] = frame512[32 * n1 + 2 * lane + 1]; }
        phase_a2(lane, v[lane], lc[lane], region.data());
    }
    for (int lane = 0; lane < 16; ++lane) phase_b1(lane, region.data(), v[lane]);
    Power p[16];
    for (int lane = 0; lane < 16; ++lane) {
        float qr[8], qi[8];
        for (int j = 0; j < 8; ++j) { int pl, pr; partner_of(lane, j, pl, pr); qr[j] = v[pl].re[pr]; qi[j] = v[pl].im[pr]; }
        phase_c1v2(lane, v[lane], qr, qi, lc[lane], p[lane]);
    }
    for (int lane = 0; lane < 16; ++lane) phase_c2(lane, p[lane], region.data());
    std::memcpy(power257, region.data(), sizeof(float) * kBins);
}
