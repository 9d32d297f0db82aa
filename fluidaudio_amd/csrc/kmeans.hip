// kmeans.hip — the speaker-count fallback of the offline diarizer on gfx950 (fp64, bit-identical to the CPU restatement).
//
// Replaces KMeansClustering.clusterWithCentroids / clusterWithCentroidsNInit (reference:
// Sources/FluidAudio/Diarizer/Offline/Clustering/KMeansClustering.swift:39-129) and SpeakerCountConstraints.resolve
// (SpeakerCountConstraints.swift:25-62), which VBxClustering.refineWithConstraints (VBxClustering.swift:685-733) calls when the
// number of clusters the VBx posteriors actually use falls outside the caller's [min, max] speakers: best of n_init = 10
// Lloyd runs (seeds 0..9, <= 100 iterations) over the unit-normalised 256-d training embeddings.
//
// Design.  The n_init runs are independent, so they are BATCHED: every kernel has a (work, run) grid and the host walks all
// runs in lock step — one stream synchronisation per Lloyd iteration for the whole batch instead of one per run (the
// reference runs them back to back).  Parity demands the reference's summation orders: distances are accumulated
// sequentially over the dimension, centroid sums sequentially over the embedding index (`sums[cluster][d] += e[d]` in index
// order, :187-193).  So
//   * assignment: one thread per embedding reads the TRANSPOSED embeddings xt[d][n] (coalesced) and carries 16 running
//     distances in registers; centroid values are wave-uniform, i.e. scalar loads.
//   * update: a stable compaction (one wave per cluster: ballot + prefix popcount) builds each cluster's member list in index
//     order, then one thread per (cluster, dimension) walks the list with 8 independent loads in flight and adds in order.
// The random draws (initial shuffle, re-seeding of empty clusters) happen on the host between iterations: they are the
// reference's LCG (SeededRNG, :212-223) pushed through the Swift standard library's `next(upperBound:)` (Lemire's method),
// `shuffle(using:)` and `randomElement(using:)`, restated here because that library is the only specification of the
// draw sequence (third-party, unpinned).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "fa_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 16;     // running distances per thread in the assignment kernel
constexpr int kMaxRuns = 64;   // runs per batch (active set is a 64-bit kernel argument)
constexpr int kAhead = 8;      // independent loads in flight per thread in the ordered centroid sums

// normalizeEmbeddings (:131-142): norm = sqrt(sum of squares); rows with norm <= 1e-10 (or NaN) are kept as they are.
__global__ void km_normalize(const double *__restrict__ x, double *__restrict__ xn, double *__restrict__ xt, int64_t n, int d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = x + i * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss = __dadd_rn(ss, __dmul_rn(r[k], r[k]));
    const double norm = __dsqrt_rn(ss);
    const bool scale = norm > 1e-10;
    const double inv = scale ? __ddiv_rn(1.0, norm) : 1.0;
    for (int k = 0; k < d; ++k) {
        const double v = scale ? __dmul_rn(r[k], inv) : r[k];
        xn[i * d + k] = v;
        xt[static_cast<int64_t>(k) * n + i] = v;
    }
}

// assignToCentroids (:154-168): first strict minimum of the sequentially accumulated squared distances.
__global__ void __launch_bounds__(kThreads) km_assign(const double *__restrict__ xt, const double *__restrict__ cen, int32_t *__restrict__ assign,
                                                      int32_t *__restrict__ changed, int64_t n, int d, int k, const int32_t *__restrict__ done) {
    const int r = blockIdx.y;
    if (done[r]) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int64_t ii = live ? i : n - 1;
    const double *c_run = cen + static_cast<int64_t>(r) * k * d;
    double bd = DBL_MAX;
    int best = 0;
    for (int c0 = 0; c0 < k; c0 += kChunk) {
        const int kk = min(kChunk, k - c0);
        double acc[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) acc[j] = 0.0;
        for (int q = 0; q < d; ++q) {
            const double x = xt[static_cast<int64_t>(q) * n + ii];
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                if (j < kk) {
                    const double df = __dsub_rn(x, c_run[static_cast<int64_t>(c0 + j) * d + q]);
                    acc[j] = __dadd_rn(acc[j], __dmul_rn(df, df));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            if (j < kk && acc[j] < bd) { bd = acc[j]; best = c0 + j; }
    }
    bool diff = false;
    if (live) {
        int32_t *a = assign + static_cast<int64_t>(r) * n + i;
        diff = *a != best;
        *a = best;
    }
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(changed + r, 1);
}

// members of cluster c of run r, counted (pass 0) or written in index order at the cluster's offset (pass 1); one wave each
template <int PASS>
__global__ void __launch_bounds__(64) km_members(const int32_t *__restrict__ assign, const int32_t *__restrict__ changed, int32_t *__restrict__ counts,
                                                 int32_t *__restrict__ list, int64_t n, int k, const int32_t *__restrict__ done) {
    const int r = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
    if (done[r] || !changed[r]) return;
    const int32_t *a = assign + static_cast<int64_t>(r) * n;
    int64_t base = 0;
    if (PASS == 1) {
        int64_t part = 0;
        for (int j = lane; j < c; j += 64) part += counts[static_cast<int64_t>(r) * k + j];
        for (int s = 32; s; s >>= 1) part += __shfl_xor(part, s);
        base = part;
    }
    int64_t total = 0;
    constexpr int kRows = 8;   // 8 x 64 assignments requested before the first is looked at (one load per step made every step a memory round trip)
    for (int64_t i0 = 0; i0 < n; i0 += 64 * kRows) {
        int32_t av[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) { const int64_t i = i0 + 64 * u + lane; av[u] = i < n ? a[i] : -1; }
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const int64_t i = i0 + 64 * u + lane;
            const bool m = av[u] == c;
            const uint64_t mask = __ballot(m);
            if (PASS == 1 && m) list[static_cast<int64_t>(r) * n + base + total + __popcll(mask & ((1ull << lane) - 1))] = static_cast<int32_t>(i);
            total += __popcll(mask);
        }
    }
    if (PASS == 0 && lane == 0) counts[static_cast<int64_t>(r) * k + c] = static_cast<int32_t>(total);
}

// updateCentroids (:179-207) for the non-empty clusters: sums in embedding-index order, times 1 / count.
__global__ void __launch_bounds__(kThreads) km_update(const double *__restrict__ xn, const int32_t *__restrict__ changed, const int32_t *__restrict__ counts,
                                                      const int32_t *__restrict__ list, double *__restrict__ cen, int64_t n, int d, int k,
                                                      const int32_t *__restrict__ done) {
    const int r = blockIdx.y, c = blockIdx.x;
    if (done[r] || !changed[r]) return;
    const int32_t *cnts = counts + static_cast<int64_t>(r) * k;
    const int cnt = cnts[c];
    if (cnt == 0) return;                       // re-seeded by km_step_end from the run's pre-drawn random stream
    int64_t base = 0;
    for (int j = 0; j < c; ++j) base += cnts[j];
    const int32_t *mine = list + static_cast<int64_t>(r) * n + base;
    const double inv = __ddiv_rn(1.0, static_cast<double>(cnt));
    for (int q = threadIdx.x; q < d; q += blockDim.x) {
        double s = 0.0;
        int j = 0;
        for (; j + kAhead <= cnt; j += kAhead) {
            double v[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) v[u] = xn[static_cast<int64_t>(mine[j + u]) * d + q];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) s = __dadd_rn(s, v[u]);
        }
        for (; j < cnt; ++j) s = __dadd_rn(s, xn[static_cast<int64_t>(mine[j]) * d + q]);
        cen[(static_cast<int64_t>(r) * k + c) * d + q] = __dmul_rn(s, inv);
    }
}

// End of a Lloyd iteration for run r (one wavefront): convergence (newAssignments == assignments, :70-73) and the re-seeding of
// empty clusters in cluster order from the run's random stream (randomElement, :196-199).  The draws do not depend on the data
// — every one is below(n) on the same generator — so the host pre-draws the sequence (picks[r][0..kPicks)) and the device only
// keeps a cursor: no host synchronisation inside the iteration loop.  status[1] is raised if a run needs more than kPicks draws.
constexpr int kPicks = 1024;
__global__ void __launch_bounds__(64) km_step_end(const int32_t *__restrict__ changed, const int32_t *__restrict__ counts, int32_t *__restrict__ done,
                                                  int32_t *__restrict__ iters, const int32_t *__restrict__ picks, int32_t *__restrict__ cursor,
                                                  const double *__restrict__ xn, double *__restrict__ cen, int32_t *__restrict__ status, int it, int d, int k) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (done[r]) return;
    if (lane == 0) iters[r] = it + 1;
    if (!changed[r]) { if (lane == 0) done[r] = 1; return; }
    int cur = cursor[r];
    for (int c = 0; c < k; ++c) {   // wave-uniform walk over the clusters, the row copy spread over the lanes
        if (counts[static_cast<int64_t>(r) * k + c] != 0) continue;
        if (cur >= kPicks) { if (lane == 0) status[1] = 1; break; }
        const int64_t pick = picks[static_cast<int64_t>(r) * kPicks + cur];
        ++cur;
        for (int q = lane; q < d; q += 64) cen[(static_cast<int64_t>(r) * k + c) * d + q] = xn[pick * d + q];
    }
    if (lane == 0) cursor[r] = cur;
}

// per-embedding squared distance to its own centroid (the terms of the inertia, :118-121)
__global__ void km_own_distance(const double *__restrict__ xt, const double *__restrict__ cen, const int32_t *__restrict__ assign,
                                double *__restrict__ dist, int64_t n, int d, int k) {
    const int r = blockIdx.y;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *c = cen + (static_cast<int64_t>(r) * k + assign[static_cast<int64_t>(r) * n + i]) * d;
    double s = 0.0;
    for (int q = 0; q < d; ++q) {
        const double df = __dsub_rn(xt[static_cast<int64_t>(q) * n + i], c[q]);
        s = __dadd_rn(s, __dmul_rn(df, df));
    }
    dist[static_cast<int64_t>(r) * n + i] = s;
}

struct Rng {   // SeededRNG (:212-223) + Swift stdlib draws
    uint64_t s;
    uint64_t next() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return s; }
    uint64_t below(uint64_t bound) {
        uint64_t r = next();
        unsigned __int128 m = static_cast<unsigned __int128>(r) * bound;
        if (static_cast<uint64_t>(m) < bound) {
            const uint64_t t = (0 - bound) % bound;
            while (static_cast<uint64_t>(m) < t) { r = next(); m = static_cast<unsigned __int128>(r) * bound; }
        }
        return static_cast<uint64_t>(m >> 64);
    }
};

struct RunResult {
    int32_t iterations = 0;
    double inertia = 0.0;
};

// Lloyd iterations for `runs` seeds at once.  On success d_assign[runs][n], d_cen[runs][k][d] hold every run's result.
fa_status lloyd_batch(fa_ctx *ctx, const double *d_xn, const double *d_xt, int64_t n, int d, int k, int max_iter, const uint64_t *seeds,
                      int runs, int32_t *d_assign, double *d_cen, int32_t *d_changed, int32_t *d_counts, int32_t *d_list,
                      std::vector<RunResult> &res) {
    hipStream_t st = ctx->stream;
    std::vector<Rng> rng(runs);
    std::vector<int64_t> idx(n);
    for (int r = 0; r < runs; ++r) {                                     // initializeCentroids (:144-152)
        rng[r].s = seeds[r];
        for (int64_t i = 0; i < n; ++i) idx[i] = i;
        int64_t amount = n, cur = 0;
        while (amount > 1) {
            const int64_t j = static_cast<int64_t>(rng[r].below(static_cast<uint64_t>(amount)));
            amount -= 1;
            std::swap(idx[cur], idx[cur + j]);
            cur += 1;
        }
        for (int c = 0; c < k; ++c)
            FA_HIP_TRY(ctx, hipMemcpyAsync(d_cen + (static_cast<int64_t>(r) * k + c) * d, d_xn + idx[c] * d, sizeof(double) * d,
                                           hipMemcpyDeviceToDevice, st));
    }
    FA_HIP_TRY(ctx, hipMemsetAsync(d_assign, 0, sizeof(int32_t) * runs * n, st));
    // pre-drawn re-seeding picks + device-side bookkeeping: done[runs] | iters[runs] | cursor[runs] | status[2] | picks[runs][kPicks]
    std::vector<int32_t> h_picks(static_cast<size_t>(runs) * kPicks);
    for (int r = 0; r < runs; ++r)
        for (int j = 0; j < kPicks; ++j) h_picks[static_cast<size_t>(r) * kPicks + j] = static_cast<int32_t>(rng[r].below(static_cast<uint64_t>(n)));
    fa::DevBuf d_book;
    const size_t book_ints = static_cast<size_t>(3) * runs + 2;
    FA_HIP_TRY(ctx, d_book.alloc(sizeof(int32_t) * (book_ints + h_picks.size())));
    int32_t *d_done = d_book.as<int32_t>(), *d_iters = d_done + runs, *d_cursor = d_iters + runs, *d_status = d_cursor + runs, *d_picks = d_status + 2;
    FA_HIP_TRY(ctx, hipMemsetAsync(d_done, 0, sizeof(int32_t) * book_ints, st));
    FA_HIP_TRY(ctx, hipMemcpyAsync(d_picks, h_picks.data(), sizeof(int32_t) * h_picks.size(), hipMemcpyHostToDevice, st));
    res.assign(runs, RunResult());
    const dim3 pgrid(static_cast<unsigned>((n + kThreads - 1) / kThreads), runs), cgrid(k, runs);
    std::vector<int32_t> h_book(book_ints);
    constexpr int kSyncEvery = 8;   // iterations between host checks (launches behind the convergence of every run are no-ops)
    for (int it = 0; it < max_iter; ++it) {
        FA_HIP_TRY(ctx, hipMemsetAsync(d_changed, 0, sizeof(int32_t) * runs, st));
        hipLaunchKernelGGL(km_assign, pgrid, dim3(kThreads), 0, st, d_xt, d_cen, d_assign, d_changed, n, d, k, d_done);
        hipLaunchKernelGGL(km_members<0>, cgrid, dim3(64), 0, st, d_assign, d_changed, d_counts, d_list, n, k, d_done);
        hipLaunchKernelGGL(km_members<1>, cgrid, dim3(64), 0, st, d_assign, d_changed, d_counts, d_list, n, k, d_done);
        hipLaunchKernelGGL(km_update, cgrid, dim3(std::min(kThreads, ((d + 63) / 64) * 64)), 0, st, d_xn, d_changed, d_counts, d_list, d_cen, n, d, k, d_done);
        hipLaunchKernelGGL(km_step_end, dim3(runs), dim3(64), 0, st, d_changed, d_counts, d_done, d_iters, d_picks, d_cursor, d_xn, d_cen, d_status, it, d, k);
        FA_HIP_TRY(ctx, hipGetLastError());
        if ((it + 1) % kSyncEvery == 0 || it + 1 == max_iter) {
            FA_HIP_TRY(ctx, hipMemcpyAsync(h_book.data(), d_done, sizeof(int32_t) * book_ints, hipMemcpyDeviceToHost, st));
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (h_book[3 * runs + 1]) return fa::set_error(ctx, FA_RUNTIME_ERROR, "kmeans: more than %d empty-cluster re-seeds in one run", kPicks);
            bool all = true;
            for (int r = 0; r < runs; ++r) all = all && h_book[r] != 0;
            if (all) break;
        }
    }
    FA_HIP_TRY(ctx, hipMemcpyAsync(h_book.data(), d_done, sizeof(int32_t) * book_ints, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));   // also keeps h_picks alive until its upload has completed
    for (int r = 0; r < runs; ++r) res[r].iterations = h_book[runs + r];
    return FA_SUCCESS;
}

struct Buffers {
    fa::DevBuf x, xn, xt, cen, assign, changed, counts, list, dist;
};

fa_status kmeans_device(fa_ctx *ctx, const double *emb, int64_t n, int d, int k, int max_iter, const uint64_t *seeds, int runs,
                        bool want_inertia, int32_t *labels, double *centroids, int32_t *best_run, double *inertias, int32_t *iterations) {
    fa::DeviceGuard guard(ctx->device);
    hipStream_t st = ctx->stream;
    int best = 0;
    double best_inertia = DBL_MAX;
    bool have = false;
    std::vector<int32_t> best_labels;
    std::vector<double> best_cen;
    for (int r0 = 0; r0 < runs; r0 += kMaxRuns) {
        const int nr = std::min(kMaxRuns, runs - r0);
        Buffers b;
        FA_HIP_TRY(ctx, b.x.alloc(sizeof(double) * n * d));
        FA_HIP_TRY(ctx, b.xn.alloc(sizeof(double) * n * d));
        FA_HIP_TRY(ctx, b.xt.alloc(sizeof(double) * n * d));
        FA_HIP_TRY(ctx, b.cen.alloc(sizeof(double) * nr * k * d));
        FA_HIP_TRY(ctx, b.assign.alloc(sizeof(int32_t) * nr * n));
        FA_HIP_TRY(ctx, b.changed.alloc(sizeof(int32_t) * nr));
        FA_HIP_TRY(ctx, b.counts.alloc(sizeof(int32_t) * nr * k));
        FA_HIP_TRY(ctx, b.list.alloc(sizeof(int32_t) * nr * n));
        FA_HIP_TRY(ctx, hipMemcpyAsync(b.x.p, emb, sizeof(double) * n * d, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(km_normalize, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(64), 0, st, b.x.as<double>(), b.xn.as<double>(),
                           b.xt.as<double>(), n, d);
        std::vector<RunResult> res;
        FA_TRY(lloyd_batch(ctx, b.xn.as<double>(), b.xt.as<double>(), n, d, k, max_iter, seeds + r0, nr, b.assign.as<int32_t>(),
                           b.cen.as<double>(), b.changed.as<int32_t>(), b.counts.as<int32_t>(), b.list.as<int32_t>(), res));
        std::vector<double> h_dist;
        if (want_inertia) {
            FA_HIP_TRY(ctx, b.dist.alloc(sizeof(double) * nr * n));
            hipLaunchKernelGGL(km_own_distance, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads), nr), dim3(kThreads), 0, st,
                               b.xt.as<double>(), b.cen.as<double>(), b.assign.as<int32_t>(), b.dist.as<double>(), n, d, k);
            FA_HIP_TRY(ctx, hipGetLastError());
            h_dist.resize(static_cast<size_t>(nr) * n);
            FA_HIP_TRY(ctx, hipMemcpyAsync(h_dist.data(), b.dist.p, sizeof(double) * nr * n, hipMemcpyDeviceToHost, st));
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        for (int r = 0; r < nr; ++r) {
            double inertia = 0.0;
            if (want_inertia) for (int64_t i = 0; i < n; ++i) inertia += h_dist[static_cast<size_t>(r) * n + i];
            if (inertias) inertias[r0 + r] = inertia;
            if (iterations && runs == 1) *iterations = res[r].iterations;
            const bool better = !want_inertia || inertia < best_inertia;  // strict '<': the first best run wins (:122-125)
            if (better || r0 + r == 0) {                                  // run 0 doubles as the fallback of :126-128
                if (better) { best_inertia = inertia; best = r0 + r; have = true; }
                best_labels.resize(n);
                best_cen.resize(static_cast<size_t>(k) * d);
                FA_HIP_TRY(ctx, hipMemcpyAsync(best_labels.data(), b.assign.as<int32_t>() + static_cast<int64_t>(r) * n, sizeof(int32_t) * n,
                                               hipMemcpyDeviceToHost, st));
                FA_HIP_TRY(ctx, hipMemcpyAsync(best_cen.data(), b.cen.as<double>() + static_cast<int64_t>(r) * k * d, sizeof(double) * k * d,
                                               hipMemcpyDeviceToHost, st));
                FA_HIP_TRY(ctx, hipStreamSynchronize(st));
            }
        }
    }
    (void)have;
    std::copy(best_labels.begin(), best_labels.end(), labels);
    if (centroids) std::copy(best_cen.begin(), best_cen.end(), centroids);
    if (best_run) *best_run = best;
    return FA_SUCCESS;
}

// the guards of clusterWithCentroids (:46-59); returns true when the call is finished without device work
bool degenerate(const double *emb, int64_t n, int32_t d, int32_t num_clusters, int32_t *labels, double *centroids, int32_t *out_k) {
    if (out_k) *out_k = 0;
    if (n <= 0) return true;
    const int64_t k = std::min<int64_t>(num_clusters, n);
    if (d <= 0 || k <= 0) { std::fill(labels, labels + n, 0); return true; }
    if (n <= k) {
        for (int64_t i = 0; i < n; ++i) labels[i] = static_cast<int32_t>(i);
        if (centroids) std::memcpy(centroids, emb, sizeof(double) * n * d);
        if (out_k) *out_k = static_cast<int32_t>(n);
        return true;
    }
    return false;
}

}  // namespace

extern "C" {

uint64_t fa_seeded_rng_next(uint64_t *state) {
    if (!state) return 0;
    Rng r{*state};
    const uint64_t v = r.next();
    *state = r.s;
    return v;
}

uint64_t fa_seeded_rng_below(uint64_t *state, uint64_t upper_bound) {
    if (upper_bound == 0 || !state) return 0;
    Rng r{*state};
    const uint64_t v = r.below(upper_bound);
    *state = r.s;
    return v;
}

fa_status fa_kmeans_cluster(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, int32_t num_clusters, int32_t max_iterations, uint64_t seed,
                            int32_t *labels, double *centroids, int32_t *out_k, int32_t *out_iterations) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (out_iterations) *out_iterations = 0;
    if (n < 0 || (n > 0 && (!labels || (d > 0 && !emb)))) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "kmeans: bad arguments");
    if (n > INT32_MAX) return fa::set_error(ctx, FA_INDEX_OVERFLOW, "kmeans: n exceeds int32");
    return fa::no_throw(ctx, "kmeans", [&]() -> fa_status {
        if (degenerate(emb, n, d, num_clusters, labels, centroids, out_k)) return FA_SUCCESS;
        const int k = static_cast<int>(std::min<int64_t>(num_clusters, n));
        FA_TRY(kmeans_device(ctx, emb, n, d, k, max_iterations, &seed, 1, false, labels, centroids, nullptr, nullptr, out_iterations));
        if (out_k) *out_k = k;
        return FA_SUCCESS;
    });
}

fa_status fa_kmeans_cluster_ninit(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, int32_t num_clusters, int32_t max_iterations, int32_t n_init,
                                  uint64_t base_seed, int32_t *labels, double *centroids, int32_t *out_k, int32_t *best_run, double *inertias) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (best_run) *best_run = 0;
    if (!(n > num_clusters && n_init > 1))                                // guard (:106-110)
        return fa_kmeans_cluster(ctx, emb, n, d, num_clusters, max_iterations, base_seed, labels, centroids, out_k, nullptr);
    if (n < 0 || !labels || (d > 0 && !emb)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "kmeans: bad arguments");
    if (n > INT32_MAX) return fa::set_error(ctx, FA_INDEX_OVERFLOW, "kmeans: n exceeds int32");
    return fa::no_throw(ctx, "kmeans", [&]() -> fa_status {
        if (degenerate(emb, n, d, num_clusters, labels, centroids, out_k)) {  // d == 0 or k <= 0: every run returns the same labels
            if (inertias) std::fill(inertias, inertias + n_init, 0.0);
            return FA_SUCCESS;
        }
        const int k = static_cast<int>(std::min<int64_t>(num_clusters, n));
        std::vector<uint64_t> seeds(n_init);
        for (int i = 0; i < n_init; ++i) seeds[i] = base_seed + static_cast<uint64_t>(i);
        FA_TRY(kmeans_device(ctx, emb, n, d, k, max_iterations, seeds.data(), n_init, true, labels, centroids, best_run, inertias, nullptr));
        if (out_k) *out_k = k;
        return FA_SUCCESS;
    });
}

void fa_speaker_constraints_resolve(int64_t num_embeddings, const int64_t *num_speakers, const int64_t *min_speakers, const int64_t *max_speakers,
                                    int64_t out[3]) {
    if (!out) return;
    int64_t rmin = num_speakers ? *num_speakers : (min_speakers ? *min_speakers : 1);
    rmin = std::max<int64_t>(1, std::min(num_embeddings, rmin));
    int64_t rmax = num_speakers ? *num_speakers : (max_speakers ? *max_speakers : num_embeddings);
    rmax = std::max<int64_t>(1, std::min(num_embeddings, rmax));
    if (rmin > rmax) rmin = rmax;
    out[0] = rmin == rmax ? rmin : (num_speakers ? *num_speakers : -1);
    out[1] = rmin;
    out[2] = rmax;
}

}  // extern "C"
