// CPU replay of the "reference order" selection of fluidaudio_amd/csrc/ahc_reforder.h (the SAME header the HIP kernels compile): the
// scans the device runs in parallel are plain loops here (the reference's sequential sums, FastClusterWrapper.cpp:45-52,68-75,89-100),
// the heap / list / merge-order logic is the header's.  tests/test_ahc_reforder_emul.py compares the dendrogram with the reference
// build (oracle/_ref) on tie-heavy inputs: row for row, bit for bit.  Test infrastructure only.
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../fluidaudio_amd/csrc/ahc_reforder.h"

namespace {
double sqdist(const double *a, const double *b, int d) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) { const double diff = a[k] - b[k]; s += diff * diff; }
    return s;
}
}  // namespace

extern "C" int fa_reforder_emul(const double *x, int n, int d, double *z /* (n-1) x 4 */) {
    if (n < 2) return 0;
    const int total = 2 * n - 1;
    std::vector<double> cent(static_cast<size_t>(total) * d), key(2 * n, 0.0), size(total, 1.0), pa(n), pb(n), hs(n);
    std::vector<int32_t> at(n), pos(2 * n, 0), nghbr(2 * n, 0), next(2 * n + 1, 0), prev(2 * n + 1, 0);
    for (size_t i = 0; i < static_cast<size_t>(n) * d; ++i) cent[i] = x[i];
    auto P = [&](int node) { return cent.data() + static_cast<size_t>(node) * d; };
    // start-up (:1653-1678): nearest LOWER-indexed point of every point, lowest index on ties
    for (int i = 1; i < n; ++i) {
        double best = std::numeric_limits<double>::infinity();
        int arg = 0;
        for (int j = 0; j < i; ++j) { const double v = sqdist(P(i), P(j), d); if (v < best) { best = v; arg = j; } }
        key[i] = best; nghbr[i] = arg;
    }
    fa_ro::Sel s{};
    s.heap.key = key.data(); s.heap.at = at.data(); s.heap.pos = pos.data();
    s.heap.init_identity(n - 1, 1);
    s.heap.heapify();
    s.list.next = next.data(); s.list.prev = prev.data();
    s.list.init(2 * n - 1);
    s.nghbr = nghbr.data(); s.n = n; s.merges = 0; s.pair_a = pa.data(); s.pair_b = pb.data(); s.height_sq = hs.data();
    s.advance();
    while (s.op != fa_ro::RO_DONE) {
        int scanned, limit;
        if (s.op == fa_ro::RO_NEW_ROW) {
            const int created = n + s.merges - 1;
            const double ma = size[s.a], mb = size[s.b], den = ma + mb;
            for (int k = 0; k < d; ++k) P(created)[k] = (P(s.a)[k] * ma + P(s.b)[k] * mb) / den;   // :89-100
            size[created] = den;
            scanned = created; limit = created;
        } else { scanned = s.a; limit = s.a; }
        double best = std::numeric_limits<double>::infinity();
        int arg = -1;
        for (int j = s.list.first; j < limit; j = s.list.next[j]) {   // active nodes below `limit`, index order, strict <
            const double v = sqdist(P(j), P(scanned), d);
            if (arg < 0 || v < best) { best = v; arg = j; }
        }
        if (best != best) return 5;
        s.scan_result(best, arg);
    }
    std::vector<double> sz(total, 1.0);
    for (int r = 0; r < n - 1; ++r) {
        const int a = static_cast<int>(pa[r]), b = static_cast<int>(pb[r]);
        sz[n + r] = sz[a] + sz[b];
        z[4 * r] = a < b ? a : b; z[4 * r + 1] = a < b ? b : a; z[4 * r + 2] = std::sqrt(hs[r]); z[4 * r + 3] = sz[n + r];
    }
    return 0;
}
