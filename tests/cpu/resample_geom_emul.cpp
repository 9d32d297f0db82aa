// CPU emulation of the indexing of poly_rows_kernel / poly_interp_kernel (fluidaudio_amd/csrc/resample.hip) on the host-side geometry the
// library computes (csrc/resample_geom.h): every staged, read and written index is range-checked, every output of the covered range is
// written exactly once, and the values are compared by the caller with a plain one-output-at-a-time evaluation (poly_simple below =
// poly_kernel of resample.hip).  Test infrastructure: built by tests/test_resample_geom_emul.py with g++, no GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../../fluidaudio_amd/csrc/resample_geom.h"


// poly_decim_tile_kernel<DOWN> on whole tiles of the outputs [10, ...) whose inputs all exist (the host's split in fa_resample_poly_dev): every staged index
// inside the signal or clamped (-2 if a READ position was clamped or lies outside), every window read inside the buffer (-3); every output of the tiles written
// once (-4 / -5); a tile's first input 16-byte aligned (-6).  Values = the ascending-input fused multiply-adds of the kernel.
template <int DOWN>
static int decim_tiles_emulate_t(const float *x, int64_t n_in, const float *h, int64_t n_out, float *y, int64_t *m_lo, int64_t *m_hi) {
    typedef fa::DecimTile<DOWN> D;
    const int64_t m_begin = 10, m_last = (n_in - 1) / DOWN - 11;
    const int64_t avail = m_last >= m_begin ? std::min(m_last + 1, n_out) - m_begin : 0;
    int64_t tiles = avail / D::TO;
    while (tiles > 0 && ((m_begin + tiles * D::TO - 1) + 11) * DOWN + 3 > n_in - 1) --tiles;   // the host's guard: the last tile's last 16-byte piece stays inside the signal
    *m_lo = *m_hi = m_begin;
    if (tiles <= 0) return 0;
    static_assert(D::RS % 4 == 0 && (D::RS / 4) % 2 == 1, "a thread's window starts 4 x odd floats behind its neighbour's: conflict-free 16-byte LDS reads");
    static_assert(D::BUF % 4 == 0 && D::BUF >= D::SPAN, "whole pieces");
    const int64_t k_lim = n_in - 4;
    std::vector<float> buf(D::BUF);
    std::vector<char> valid(D::BUF), written(static_cast<size_t>(tiles * D::TO), 0);
    for (int64_t t = 0; t < tiles; ++t) {
        const int64_t k0 = (m_begin + t * D::TO - 10) * DOWN;
        if (k0 % 4 != 0 || k0 < 0) return -6;
        std::fill(valid.begin(), valid.end(), 0);
        const int nreq = (D::PIECES + 63) / 64;
        for (int req = 0; req < nreq; ++req)
            for (int lane = 0; lane < 64; ++lane) {
                int64_t off = 4 * (req * 64 + lane);
                const bool clamped = off > k_lim - k0;
                if (clamped) off = k_lim - k0;
                if (k0 + off < 0 || k0 + off + 3 >= n_in) return -2;
                for (int e = 0; e < 4; ++e) {
                    const size_t at = static_cast<size_t>(256 * req + 4 * lane + e);
                    if (at >= buf.size()) return -3;
                    buf[at] = x[k0 + off + e];
                    valid[at] = clamped ? 0 : 1;
                }
            }
        for (int i = 0; i < fa::kDecimThreads; ++i) {
            if (i * D::RS + 16 * D::NB > D::BUF) return -3;
            float acc[D::R];
            for (int j = 0; j < D::R; ++j) acc[j] = 0.0f;
            for (int p = 0; p < D::NIN; ++p) {
                if (!valid[static_cast<size_t>(i * D::RS + p)]) return -2;
                for (int j = 0; j < D::R; ++j) {
                    const int ti = D::NT - 1 + j * DOWN - p;
                    if (ti >= 0 && ti < D::NT) acc[j] = fmaf(h[ti], buf[static_cast<size_t>(i * D::RS + p)], acc[j]);
                }
            }
            for (int j = 0; j < D::R; ++j) {
                const int64_t m = m_begin + t * D::TO + D::R * i + j;
                if (m >= n_out || written[static_cast<size_t>(m - m_begin)]) return -4;
                written[static_cast<size_t>(m - m_begin)] = 1;
                y[m] = acc[j];
            }
        }
    }
    for (char c : written) if (!c) return -5;
    *m_hi = m_begin + tiles * D::TO;
    return 0;
}

extern "C" {

// poly_kernel: y[m] for m in [m_lo, m_hi)
void poly_simple(const float *x, int64_t n_in, const float *h, int64_t h_len, int up, int down, int64_t pre_remove, float *y, int64_t m_lo, int64_t m_hi) {
    for (int64_t m = m_lo; m < m_hi; ++m) {
        const int64_t p = (m + pre_remove) * down;
        int64_t k_hi = p / up, k_lo = (p - (h_len - 1) + up - 1) / up;
        if (p - (h_len - 1) < 0) k_lo = 0;
        if (k_hi > n_in - 1) k_hi = n_in - 1;
        float acc = 0.0f;
        for (int64_t k = k_lo; k <= k_hi; ++k) acc = fmaf(h[p - k * up], x[k], acc);
        y[m] = acc;
    }
}

// returns 0 and the covered range, or a negative code: -1 pair not served, -2 staged index out of the signal, -3 window read outside the staged
// span, -4 an output written twice / outside [m_begin, m_stop), -5 an output of the range not written, -6 misaligned window read,
// -7 the phases of a shared window disagree about it, -8 a tap at a window position the kernel does not multiply
int rows_emulate(const float *x, int64_t n_in, const float *h, int64_t h_len, int up, int down, int64_t pre_remove, int64_t n_out, float *y, int64_t *m_lo, int64_t *m_hi,
                 int32_t *info /* nv, groups, ppg, sld, tiles, share */, int share_max, int64_t lds_budget, int rows) {
    fa::PolyRowsGeom g{};
    int nv = 0;
    std::vector<int> gtab;
    std::vector<float> tt;
    std::vector<float> hv(h, h + h_len);
    *m_lo = *m_hi = 0;
    if (!fa::rows_geometry(g, nv, hv, up, down, pre_remove, gtab, tt, static_cast<size_t>(lds_budget), share_max)) return -1;
    const int64_t tiles = fa::rows_tiles(g, n_in, n_out, rows);
    info[0] = nv; info[1] = g.groups; info[2] = g.ppg; info[3] = g.sld; info[4] = static_cast<int32_t>(tiles); info[5] = g.share;
    if (tiles <= 0) return 0;
    const int64_t m_stop = std::min<int64_t>(n_out, g.m_begin + tiles * rows * static_cast<int64_t>(g.up));
    std::vector<char> written(static_cast<size_t>(m_stop - g.m_begin), 0);
    std::vector<float> xs(static_cast<size_t>(rows) * g.sld);
    std::vector<char> staged(xs.size());
    const int NT = std::min(4 * nv, fa::kRowsOffLane);
    std::vector<std::pair<int64_t, int>> order;
    for (int64_t tile = 0; tile < tiles; ++tile) for (int grp = 0; grp < g.groups; ++grp) order.emplace_back(tile, grp);
    for (const auto &item : order) {
            const int64_t tile = item.first;
            const int grp = item.second;
            {
            const int ph0 = grp * g.ppg, ph1 = std::min(ph0 + g.ppg, g.up);
            const int smin = gtab[2 * grp], span = gtab[2 * grp + 1];
            if (span > g.sld || (smin & 3)) return -3;
            std::fill(staged.begin(), staged.end(), 0);
            const int64_t kt = g.k_begin + tile * rows * g.down + smin;
            for (int l = 0; l < rows; ++l)
                for (int sidx = 0; sidx < span; ++sidx) {
                    const int64_t k = kt + static_cast<int64_t>(l) * g.down + sidx;
                    if (k < 0 || k >= n_in) return -2;
                    xs[static_cast<size_t>(l) * g.sld + sidx] = x[k];
                    staged[static_cast<size_t>(l) * g.sld + sidx] = 1;
                }
            for (int ph = ph0; ph < ph1; ++ph) {
                const float *row = tt.data() + static_cast<size_t>(ph) * fa::kRowsTT;
                // the phase reads the window of its share-aligned sub-chunk (the kernel fetches the offset from the sub-chunk's FIRST phase)
                const int qph = ph0 + (ph - ph0) / g.share * g.share;
                if (qph % g.share != 0) return -7;                          // groups start at multiples of 4 >= share: local and global alignment agree
                int off4, off4_own;
                memcpy(&off4, tt.data() + static_cast<size_t>(qph) * fa::kRowsTT + fa::kRowsOffPos, sizeof(int));
                memcpy(&off4_own, row + fa::kRowsOffPos, sizeof(int));
                if (off4 != off4_own) return -7;                            // every phase of a shared window carries the same offset
                if (off4 & 3) return -6;
                for (int j = NT; j < fa::kRowsOffPos; ++j) if (row[j] != 0.0f) return -8;   // no tap beyond the positions the kernel multiplies
                for (int l = 0; l < rows; ++l) {
                    const int base = l * g.sld + (off4 - smin);
                    if (off4 - smin < 0 || off4 - smin + 4 * nv > g.sld) return -3;
                    float acc = 0.0f;
                    for (int v = 0; v < 4 * nv; ++v) if (!staged[static_cast<size_t>(base) + v]) return -3;   // the 16-byte reads touch staged data only
                    for (int j = 0; j < NT; ++j) acc = fmaf(row[j], xs[static_cast<size_t>(base) + j], acc);
                    const int64_t m = g.m_begin + tile * rows * g.up + static_cast<int64_t>(l) * g.up + ph;
                    if (m < m_stop) {
                        if (m < g.m_begin || written[static_cast<size_t>(m - g.m_begin)]) return -4;
                        written[static_cast<size_t>(m - g.m_begin)] = 1;
                        y[m] = acc;
                    }
                }
            }
            }
        }
    for (char c : written) if (!c) return -5;
    *m_lo = g.m_begin; *m_hi = m_stop;
    return 0;
}

// poly_interp_kernel<up, down, nt>, R = 4
int interp_emulate(const float *x, int64_t n_in, const float *h, int nt, int up, int down, int64_t pre_remove, int64_t n_out, float *y, int64_t *m_lo, int64_t *m_hi) {
    const int R = 4, NO = R * up, KB = (nt - 1) / up, NIN = ((NO - 1) * down) / up + KB + 1, NV = (NIN + 3) / 4;
    int64_t m_begin = 0, q_begin = 0, groups = 0;
    fa::interp_geometry(up, down, nt, R, n_in, n_out, pre_remove, m_begin, q_begin, groups);
    *m_lo = *m_hi = 0;
    if (groups <= 0) return 0;
    if (((m_begin + pre_remove) * down) % up != 0) return -6;          // a thread's first output starts a phase cycle
    for (int64_t g = 0; g < groups; ++g) {
        const int64_t k0 = q_begin - KB + g * (R * down);
        if (k0 < 0 || k0 + 4 * NV > n_in) return -2;                   // the 16-byte loads stay inside the signal
        for (int r = 0; r < NO; ++r) {
            const int64_t m = m_begin + g * NO + r;
            if (m >= n_out) return -4;
            float acc = 0.0f;
            for (int i = 0; i < NIN; ++i) {
                const int ti = r * down + (KB - i) * up;
                if (ti >= 0 && ti < nt) acc = fmaf(h[ti], x[k0 + i], acc);
            }
            y[m] = acc;
        }
    }
    *m_lo = m_begin; *m_hi = m_begin + groups * NO;
    return 0;
}

int decim_tiles_emulate(const float *x, int64_t n_in, const float *h, int down, int64_t n_out, float *y, int64_t *m_lo, int64_t *m_hi) {
    switch (down) {
        case 2: return decim_tiles_emulate_t<2>(x, n_in, h, n_out, y, m_lo, m_hi);
        case 3: return decim_tiles_emulate_t<3>(x, n_in, h, n_out, y, m_lo, m_hi);
        case 4: return decim_tiles_emulate_t<4>(x, n_in, h, n_out, y, m_lo, m_hi);
        case 5: return decim_tiles_emulate_t<5>(x, n_in, h, n_out, y, m_lo, m_hi);
        case 6: return decim_tiles_emulate_t<6>(x, n_in, h, n_out, y, m_lo, m_hi);
        case 12: return decim_tiles_emulate_t<12>(x, n_in, h, n_out, y, m_lo, m_hi);
        default: return -1;
    }
}

}  // extern "C"
