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
