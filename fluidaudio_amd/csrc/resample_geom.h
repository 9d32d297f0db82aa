// resample_geom.h — host-side geometry of the polyphase kernels of resample.hip, in plain C++ so that the CPU tests can run it
// (tests/cpu/resample_geom_emul.cpp emulates the kernels' indexing with it and checks every staged / read / written index).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace fa {

struct PolyRowsGeom {
    int64_t m_begin, k_begin;       // first output (a multiple of 4) / first staged input of tile 0
    int32_t up, down;
    int32_t groups, ppg;            // phase groups per tile, phases per group (a multiple of 4)
    int32_t sld;                    // LDS row stride in floats (4 x odd)
    int32_t smax;                   // last staged offset + 1 within a row over all phases (for the tile-count bound)
    int32_t share;                  // consecutive phases that read ONE register window (1, 2 or 4; round 5)
};
constexpr int kRowsThreads = 512, kRowsWaves = kRowsThreads / 64;
constexpr int kRowsOffLane = 128;  // window positions a table row can carry taps for: 0 .. 63 = one window of 16 x 16-byte reads (four table registers of 16
                                   // taps), 64 .. 127 = a second window right behind it for phases of more than 64 taps (88.2 -> 16 kHz: 116 taps per phase)
constexpr int kRowsOffPos = 128;   // position of a table row that carries the aligned window offset
constexpr int kRowsTT = 144;       // floats per table row (a multiple of 16: the 64-byte table pieces stay aligned)

// poly_rows_kernel: tables of one rate pair.  h = the FIR of fa_resample_poly_taps (leading zero taps included), gtab = {first staged offset,
// staged span} per phase group, tt = kRowsTT floats per phase, nv = 16-byte reads per register window.
// Round 5: `share` consecutive phases (aligned to `share` within a group) read ONE window of 4 nv floats from the lane's LDS row: their input windows
// overlap almost completely (consecutive phases start down / up = 2.76 samples apart at 44.1 -> 16 kHz and hold 56 taps), so the round-4 form —
// every phase its own 16 x 16-byte LDS reads — read each staged sample ~20 times and was LDS-bandwidth bound next to its arithmetic.  tt[ph][i] =
// the tap that multiplies window position i of the phase's SHARED window (zeros in front and behind: the shift of the phase inside the window
// is absorbed by the table, so every multiply-add has compile-time register indices); tt[ph][128] = the window's first staged offset (a
// multiple of 4, the same for the phases that share it).  false: the pair does not suit the kernel.
inline bool rows_geometry(PolyRowsGeom &g, int &nv_out, const std::vector<float> &h, int up, int down, int64_t pre_remove, std::vector<int> &gtab, std::vector<float> &tt,
                          size_t lds_budget = 0, int share_max = 4) {
    const int64_t h_len = static_cast<int64_t>(h.size());
    if (up < 8 || up > 4096 || down > 8192) return false;               // few phases: the register-tiled kernels; huge ones: tables too large
    const int q1 = static_cast<int>((h_len + up - 1) / up);             // a window holds floor(h_len / up) or that + 1 taps
    if (q1 + 3 > kRowsOffLane) return false;                            // the shifted taps of a phase must fit the 128 positions of a table row
    // first output whose window lies inside the signal: p - (h_len - 1) >= 0; rounded up to a multiple of 4 (16-byte output pieces)
    int64_t m_begin = (h_len - 1 + down - 1) / down - pre_remove;
    if (m_begin < 0) m_begin = 0;
    m_begin = (m_begin + 3) & ~static_cast<int64_t>(3);
    const int64_t p0 = (m_begin + pre_remove) * down;
    const int64_t k_begin = (p0 - (h_len - 1) + up - 1) / up;           // k_lo of the first output
    std::vector<int> off(up), cnt(up);
    for (int ph = 0; ph < up; ++ph) {
        const int64_t p = p0 + static_cast<int64_t>(ph) * down;
        const int64_t k_hi = p / up, k_lo = (p - (h_len - 1) + up - 1) / up;   // p - (h_len - 1) >= 0 here
        cnt[ph] = static_cast<int>(k_hi - k_lo + 1);
        off[ph] = static_cast<int>(k_lo - k_begin);
        if (cnt[ph] < 1) return false;
    }
    static const int sizes[] = {4, 6, 8, 10, 12, 14, 16};               // instantiated window sizes (poly_rows_launch); a larger one only reads a little further into the row
    // the reach of a shared window: phases q .. q + share - 1 (q a multiple of `share`; groups start at multiples of 4) measured from off[q] & ~3
    auto reach_of = [&](const int share) {
        int reach = 0;
        for (int q = 0; q < up; q += share) {
            const int base4 = off[q] & ~3;
            for (int u = 0; u < share && q + u < up; ++u) reach = std::max(reach, off[q + u] - base4 + cnt[q + u]);
        }
        return reach;
    };
    int share = 1, nv = 0;
    for (int c : {4, 2, 1}) {
        if (c > share_max) continue;
        const int reach = reach_of(c);
        if (reach > kRowsOffLane) continue;
        int pick = 0;
        for (int v : sizes) if (4 * v >= reach) { pick = v; break; }
        if (!pick && c == 1 && reach <= 128) pick = 32;                 // more than 64 taps per phase: two windows of 16 reads, one behind the other (no sharing)
        if (!pick) continue;
        share = c; nv = pick;
        break;
    }
    if (!nv) return false;
    // phase groups (a multiple of 4 phases each): the rows of a group within the LDS budget; rows are `sld` floats apart, sld = 4 x odd >= the longest
    // staged span of a group.  Budget (0 = automatic): short windows (<= 8 reads per window) get 38 KB = three to four workgroups per CU staging and
    // computing side by side; long windows keep 74 KB = two per CU (profiles/r04_rows_lds_probe.json, r04_rows_lds_ab.json)
    if (lds_budget == 0) lds_budget = nv <= 8 ? 38 * 1024 : 74 * 1024;
    auto span_of = [&](const int a0, const int a1) {     // staged span of the phases [a0, a1): from the first window's start to the end of the last window read
        int hi = 0;
        for (int q = a0; q < a1; q += share) hi = std::max(hi, (off[q] & ~3) + 4 * nv);
        return hi - (off[a0] & ~3);
    };
    int groups = 1, ppg = up, sld = 0;
    for (;; ++groups) {
        ppg = ((up + groups - 1) / groups + 3) & ~3;
        int span = 0;
        for (int a0 = 0; a0 < up; a0 += ppg) span = std::max(span, span_of(a0, std::min(up, a0 + ppg)));
        sld = (span + 3) / 4;
        if (sld % 2 == 0) ++sld;
        sld *= 4;
        if (static_cast<size_t>(sld) * 64 * sizeof(float) <= lds_budget || ppg <= 4 * kRowsWaves) break;
    }
    groups = (up + ppg - 1) / ppg;
    if (static_cast<size_t>(sld) * 64 * sizeof(float) > 150 * 1024 || sld > 64 * 10) return false;
    gtab.assign(2 * static_cast<size_t>(groups), 0);
    int smax = 0;
    for (int gq = 0; gq < groups; ++gq) {
        const int a0 = gq * ppg, a1 = std::min(up, a0 + ppg);
        gtab[2 * gq] = off[a0] & ~3;
        gtab[2 * gq + 1] = span_of(a0, a1);
        smax = std::max(smax, gtab[2 * gq] + gtab[2 * gq + 1]);
    }
    tt.assign(static_cast<size_t>(up + 16) * kRowsTT, 0.0f);             // empty rows behind the last phase: poly_rows_wide_kernel fetches units of 8 / 16 phases without clamps
    for (int ph = 0; ph < up; ++ph) {
        const int q = ph - ph % share;                                  // (groups start at multiples of 4: a shared window never straddles two groups)
        const int base4 = off[q] & ~3, a = off[ph] - base4;
        const int64_t p = p0 + static_cast<int64_t>(ph) * down;
        const int64_t k_lo = k_begin + off[ph];
        float *rowp = tt.data() + static_cast<size_t>(ph) * kRowsTT;
        for (int j = 0; j < cnt[ph]; ++j) rowp[a + j] = h[static_cast<size_t>(p - (k_lo + j) * up)];
        memcpy(rowp + kRowsOffPos, &base4, sizeof(int));
    }
    g.m_begin = m_begin; g.k_begin = k_begin; g.up = up; g.down = down; g.groups = groups; g.ppg = ppg; g.sld = sld; g.smax = smax; g.share = share;
    nv_out = nv;
    return true;
}

// tiles of `rows` x up outputs whose staged inputs all exist: tile t stages x[k_begin + rows down t ... + (rows - 1) down + smax)
// (rows = 64: poly_rows_kernel; 32: poly_rows_wide_kernel)
inline int64_t rows_tiles(const PolyRowsGeom &G, int64_t frames, int64_t n_out, int rows = 64) {
    const int64_t last_need = G.k_begin + static_cast<int64_t>(rows - 1) * G.down + G.smax;      // exclusive end of tile 0's staged range
    int64_t tiles = frames >= last_need ? (frames - last_need) / (static_cast<int64_t>(rows) * G.down) + 1 : 0;
    const int64_t per_tile = static_cast<int64_t>(rows) * G.up;
    if (G.m_begin < n_out) tiles = std::min(tiles, (n_out - G.m_begin + per_tile - 1) / per_tile); else tiles = 0;
    return tiles;
}

// poly_interp_kernel<UP, DOWN, NT> with R outputs-per-phase-cycle groups per thread: first output, first phase cycle, number of threads' groups
inline void interp_geometry(int up, int down, int nt, int r, int64_t frames, int64_t n_out, int64_t pre_remove, int64_t &m_begin, int64_t &q_begin, int64_t &groups) {
    const int no = r * up, kb = (nt - 1) / up, nin = ((no - 1) * down) / up + kb + 1, nv = (nin + 3) / 4;
    // first output: (m + pre_remove) = j UP with j DOWN >= KB (the first input of the thread exists)
    int64_t j = (kb + down - 1) / down;
    while (j * up < pre_remove) ++j;
    m_begin = j * up - pre_remove; q_begin = j * down;
    // group g reads x[q_begin - KB + g R DOWN ... + 4 NV): inside the signal; its outputs below n_out
    groups = 0;
    const int64_t first = q_begin - kb;
    if (frames >= first + 4 * nv) groups = (frames - first - 4 * nv) / (static_cast<int64_t>(r) * down) + 1;
    if (m_begin < n_out) groups = std::min(groups, (n_out - m_begin) / no); else groups = 0;
}

// poly_decim_tile_kernel<DOWN>: the constants of a tile (resample.hip explains the kernel)
constexpr int kDecimThreads = 256;
template <int DOWN>
struct DecimTile {
    static constexpr int R = DOWN == 2 ? 6 : DOWN == 3 ? 4 : DOWN == 4 ? 3 : DOWN == 5 ? 4 : DOWN == 6 ? 2 : 1;     // R DOWN = 12, 12, 12, 20, 12 floats between two threads' windows; DOWN = 12 (192 kHz): R = 1
    static constexpr int NT = 21 * DOWN + 1, RS = R * DOWN, NIN = NT + (R - 1) * DOWN, NB = (NIN + 15) / 16;
    static constexpr int TO = kDecimThreads * R;                                  // outputs of a tile
    static constexpr int SPAN = TO * DOWN + NT - 1, PIECES = (SPAN + 3) / 4;   // its inputs, in floats and in 16-byte pieces
    static constexpr int BUF_A = (PIECES + 63) / 64 * 64 * 4, BUF_B = (kDecimThreads - 1) * RS + 16 * NB;
    static constexpr int BUF = BUF_A > BUF_B ? BUF_A : BUF_B;                // floats per buffer: whole requests, and the last thread's last (partly unused) quarter
};

}  // namespace fa
