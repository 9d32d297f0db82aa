// CPU replay of the matrix-filtered reference-order run (fluidaudio_amd/csrc/ahc_rom.hip: rom_scan / rom_select) on the SAME header the HIP kernels
// compile (ahc_reforder.h: HeapK — entries that carry their key, block-wise sifts — and SelT).  Two entry points:
//   fa_heapk_equiv : Heap (the statement-for-statement restatement of the reference's binary_min_heap) and HeapK side by side on a random
//                    stream of remove / replace / raise with heavily tied keys; every array compared after every operation.
//   fa_rom_emul    : the whole dendrogram the way the device computes it — a Lance-Williams matrix (Gram-form start, kept symmetric
//                    over the live slots, the device's eps) supplies the candidates of every scan — the start-up scans included — (entries within 2 eps of the approximate minimum), the few
//                    candidates are evaluated with the reference's sequential sums, the heap replay picks the pair.  `noise` perturbs the
//                    start-up matrix by up to noise * eps per entry (a stand-in for the worst rounding the bound allows).
// tests/test_ahc_reforder_emul.py compares the result with the reference build (oracle/_ref) row for row.  Test infrastructure only.
#include <cmath>
#include <cstdint>
#include <limits>
#include <random>
#include <vector>

#include "../../fluidaudio_amd/csrc/ahc_reforder.h"

namespace {
double sqdist(const double *a, const double *b, int d) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) { const double diff = a[k] - b[k]; s += diff * diff; }
    return s;
}
using HeapS = fa_ro::HeapK<fa_ro::SerialMem>;
}  // namespace

extern "C" int fa_heapk_equiv(int n, int ops, unsigned long long seed, int key_levels) {
    std::mt19937_64 rng(seed);
    auto draw = [&]() { return static_cast<double>(rng() % static_cast<unsigned long long>(key_levels)); };
    const int cap = 2 * n + ops + 8;
    std::vector<double> key(cap, 0.0);
    std::vector<int32_t> at(n), pos(cap, -1), posk(cap, -1);
    std::vector<fa_ro::Ent> ent(n);
    for (int i = 0; i < n; ++i) key[i] = draw();
    fa_ro::Heap h{};
    h.key = key.data(); h.at = at.data(); h.pos = pos.data();
    h.init_identity(n, 0);
    h.heapify();
    HeapS k{};
    k.ent = ent.data(); k.pos = posk.data(); k.size = n;
    for (int p = 0; p < n; ++p) { ent[p].key = key[at[p]]; ent[p].node = at[p]; ent[p].pad = 0; posk[at[p]] = p; }
    auto same = [&]() {
        if (h.size != k.size) return false;
        for (int p = 0; p < h.size; ++p) {
            if (at[p] != ent[p].node || key[at[p]] != ent[p].key) return false;
            if (pos[at[p]] != p || posk[at[p]] != p) return false;
        }
        return h.size == 0 || (h.argmin() == k.argmin() && h.top_key() == k.top_key());
    };
    if (!same()) return -1;
    int next_node = n;
    for (int o = 0; o < ops && h.size > 1; ++o) {
        const int place = static_cast<int>(rng() % static_cast<unsigned long long>(h.size));
        const int node = at[place];
        const int what = static_cast<int>(rng() % 3);
        if (what == 0) { h.remove(node); k.remove(node); }
        else if (what == 1) { const double v = draw(); const int nn = next_node++; h.replace(node, nn, v); k.replace(node, nn, v); }
        else { const double v = key[node] + static_cast<double>(rng() % 3); h.raise(node, v); k.raise(node, v); }
        if (!same()) return o + 1;
    }
    return 0;
}

// stats: [0] scans, [1] candidates evaluated, [2] largest candidate set, [3] eps
extern "C" int fa_rom_emul(const double *x, int n, int d, double noise, unsigned long long seed, double *z /* (n-1) x 4 */, double *stats) {
    if (n < 2) return 0;
    const int total = 2 * n - 1;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> cent(static_cast<size_t>(total) * d), size(total, 1.0), pa(n), pb(n), hs(n), key0(2 * n, 0.0);
    std::vector<int32_t> at(n), pos(2 * n, 0), nghbr(2 * n, 0), next(2 * n + 1, 0), prev(2 * n + 1, 0);
    for (size_t i = 0; i < static_cast<size_t>(n) * d; ++i) cent[i] = x[i];
    auto P = [&](int node) { return cent.data() + static_cast<size_t>(node) * d; };
    // the filter: Gram-form matrix over slots (slot i = point i), eps as ahc_set_eps + the term of the sequentially summed d(a, b)
    std::vector<double> M(static_cast<size_t>(n) * n, inf), norm(n, 0.0);
    std::vector<int32_t> node_of(n), slot_of(total, -1);
    double dmax = 0.0, nmax = 0.0;
    for (int i = 0; i < n; ++i) { node_of[i] = i; slot_of[i] = i; for (int k = 0; k < d; ++k) norm[i] += P(i)[k] * P(i)[k]; if (norm[i] > nmax) nmax = norm[i]; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            double dot = 0.0;
            for (int k = 0; k < d; ++k) dot += P(i)[k] * P(j)[k];
            double v = norm[i] + norm[j] - 2.0 * dot;
            if (!(v > 0.0)) v = 0.0;
            if (v > dmax) dmax = v;
            M[static_cast<size_t>(i) * n + j] = M[static_cast<size_t>(j) * n + i] = v;
        }
    const double u = 1.1102230246251565e-16;
    const double eps = (16.0 + 0.25 * (d + 2.0)) * n * u * dmax + 8.0 * (d + 2.0) * u * nmax;
    if (noise > 0.0) {
        std::mt19937_64 rng(seed);
        std::uniform_real_distribution<double> un(-1.0, 1.0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) {
                double v = M[static_cast<size_t>(i) * n + j] + noise * eps * un(rng);
                if (!(v > 0.0)) v = 0.0;
                M[static_cast<size_t>(i) * n + j] = M[static_cast<size_t>(j) * n + i] = v;
            }
    }
    // start-up (:1653-1678) the way rom_lower_minima computes it: the nearest LOWER-indexed neighbour of every point from the (perturbed) matrix row —
    // smallest entry left of the diagonal, every entry within 2 eps of it a candidate, the reference's sum for those, lowest (value, index)
    double start_cands = 0;
    for (int i = 1; i < n; ++i) {
        double m = inf;
        for (int j = 0; j < i; ++j) if (M[static_cast<size_t>(i) * n + j] < m) m = M[static_cast<size_t>(i) * n + j];
        const double lim = m + 2.0 * eps;
        double best = inf;
        int arg = 0;
        for (int j = 0; j < i; ++j) {
            if (!(M[static_cast<size_t>(i) * n + j] <= lim)) continue;
            const double v = sqdist(P(i), P(j), d);
            if (v != v) return 5;
            ++start_cands;
            if (v < best) { best = v; arg = j; }
        }
        key0[i] = best; nghbr[i] = arg;
    }
    // the initial heap through the restated heap, then entry form (what the device's host side does)
    fa_ro::Heap h0{};
    h0.key = key0.data(); h0.at = at.data(); h0.pos = pos.data();
    h0.init_identity(n - 1, 1);
    h0.heapify();
    std::vector<fa_ro::Ent> ent(n);
    for (int p = 0; p < h0.size; ++p) { ent[p].key = key0[at[p]]; ent[p].node = at[p]; ent[p].pad = 0; }
    fa_ro::SelT<HeapS> s{};
    s.heap.ent = ent.data(); s.heap.pos = pos.data(); s.heap.size = h0.size;
    s.list.next = next.data(); s.list.prev = prev.data();
    s.list.init(2 * n - 1);
    s.nghbr = nghbr.data(); s.n = n; s.merges = 0; s.pair_a = pa.data(); s.pair_b = pb.data(); s.height_sq = hs.data();
    // the device keeps the matrix symmetric over the live slots (the column of every new row is written too), so an entry is always read as a row copy
    auto entry = [&](int r, int, int c, int) { return M[static_cast<size_t>(r) * n + c]; };
    double n_scans = 0, n_cand = 0, max_cand = 0;
    s.advance();
    std::vector<double> vals(n);
    while (s.op != fa_ro::RO_DONE) {
        int scanned, limit, ss;
        if (s.op == fa_ro::RO_NEW_ROW) {
            const int created = n + s.merges - 1, sa = slot_of[s.a], sb = slot_of[s.b];
            const double ma = size[s.a], mb = size[s.b], den = ma + mb;
            for (int k = 0; k < d; ++k) P(created)[k] = (P(s.a)[k] * ma + P(s.b)[k] * mb) / den;   // :89-100
            size[created] = den;
            const double dab = hs[s.merges - 1];                     // the exact squared height of the pair (its heap key)
            const double inv = 1.0 / den, wa = ma * inv, wb = mb * inv, wab = wa * wb;
            for (int c = 0; c < n; ++c) {
                vals[c] = inf;
                if (node_of[c] < 0 || c == sa || c == sb) continue;
                double v = wa * entry(sa, s.a, c, node_of[c]) + wb * entry(sb, s.b, c, node_of[c]) - wab * dab;
                if (!(v > 0.0)) v = 0.0;
                vals[c] = v;
            }
            for (int c = 0; c < n; ++c) if (vals[c] < inf) M[static_cast<size_t>(sa) * n + c] = M[static_cast<size_t>(c) * n + sa] = vals[c];   // row (rom_scan) and column (the mirror workgroups of rom_select)
            node_of[sa] = created; slot_of[created] = sa; node_of[sb] = -1;
            scanned = created; limit = created; ss = sa;
        } else {
            scanned = s.a; limit = s.a; ss = slot_of[s.a];
            for (int c = 0; c < n; ++c) vals[c] = (node_of[c] >= 0 && c != ss && node_of[c] < limit) ? M[static_cast<size_t>(ss) * n + c] : inf;
        }
        double m = inf;
        for (int c = 0; c < n; ++c) if (vals[c] < m) m = vals[c];
        if (!(m < inf)) return 6;                                      // nothing to scan: the reference never asks for that
        const double lim = m + 2.0 * eps;
        double best = inf, cands = 0;
        int arg = std::numeric_limits<int>::max();
        for (int c = 0; c < n; ++c) {
            if (!(vals[c] <= lim)) continue;
            const double v = sqdist(P(node_of[c]), P(scanned), d);      // the reference's sum (either operand order: the square is the same)
            if (v != v) return 5;
            ++cands;
            if (v < best || (v == best && node_of[c] < arg)) { best = v; arg = node_of[c]; }
        }
        ++n_scans; n_cand += cands; if (cands > max_cand) max_cand = cands;
        s.scan_result(best, arg);
    }
    std::vector<double> sz(total, 1.0);
    for (int r = 0; r < n - 1; ++r) {
        const int a = static_cast<int>(pa[r]), b = static_cast<int>(pb[r]);
        sz[n + r] = sz[a] + sz[b];
        z[4 * r] = a < b ? a : b; z[4 * r + 1] = a < b ? b : a; z[4 * r + 2] = std::sqrt(hs[r]); z[4 * r + 3] = sz[n + r];
    }
    if (stats) { stats[0] = n_scans; stats[1] = n_cand + start_cands; stats[2] = max_cand; stats[3] = eps; }
    return 0;
}
