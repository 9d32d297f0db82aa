// ahc_api.hip — the C entries of the linkage, normalisation, the dendrogram cut, the shardable nearest-neighbour table (ahc_ws.h: the map).
#include "ahc_ws.h"

using namespace fa_ahc;

namespace {

// normalizeFeatures (AHCClustering.swift:70-105): one thread per row, the reference's sequential sum of squares (this file
// is compiled with -ffp-contract=off), scale = norm > 0 ? 1 / sqrt(norm) : 0 — bit-identical to the host loop of fa_ahc_cluster.
__global__ void ahc_normalize_rows(const double *__restrict__ x, double *__restrict__ out, int64_t n, int d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *row = x + i * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss += row[k] * row[k];
    const double scale = ss > 0 ? 1.0 / sqrt(ss) : 0.0;
    for (int k = 0; k < d; ++k) out[i * d + k] = row[k] * scale;
}

}  // namespace

fa_status fa::ahc_normalize_dev(fa_ctx *ctx, const double *d_x, double *d_out, int64_t n, int32_t d) {
    if (n <= 0) return FA_SUCCESS;
    hipLaunchKernelGGL(ahc_normalize_rows, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(64), 0, ctx->stream, d_x, d_out, n, d);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

namespace {

fa_status linkage_checks(const double *data, size_t n, size_t d, double *z, size_t zlen, bool *trivial) {
    // status contract of FastClusterWrapper.cpp:203-226
    *trivial = true;
    if (!data || !z) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (d == 0) return FA_INVALID_ARGUMENT;
    if (n > static_cast<size_t>(INT32_MAX) || d > static_cast<size_t>(INT32_MAX)) return FA_INDEX_OVERFLOW;
    const size_t need = n > 1 ? (n - 1) * 4 : 0;
    if (zlen < need) return FA_OUTPUT_TOO_SMALL;
    if (n == 1) return FA_SUCCESS;
    *trivial = false;
    return FA_SUCCESS;
}

}  // namespace

extern "C" {

fa_status fa_ahc_linkage(fa_ctx *ctx, const double *data, size_t n, size_t d, double *dendrogram, size_t dendrogram_len,
                         int32_t mode, int32_t device_pointers, fa_ahc_stats *stats) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    bool trivial;
    const fa_status pre = linkage_checks(data, n, d, dendrogram, dendrogram_len, &trivial);
    if (pre != FA_SUCCESS || trivial) return pre;
    if (stats) memset(stats, 0, sizeof(*stats));
    try {
        fa::DeviceGuard guard(ctx->device);
        if (device_pointers) return fa::ahc_run_device(ctx, data, n, d, dendrogram, mode, stats);
        // input staged in the context's grow-only scratch (no hipMalloc / hipFree per call), dendrogram copied straight from the workspace
        FA_TRY(fa::ensure_scratch(ctx, sizeof(double) * n * d));
        FA_HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch, data, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream));
        return fa::ahc_run_device(ctx, static_cast<const double *>(ctx->scratch), n, d, dendrogram, mode, stats, /*z_on_host*/ true);
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FA_RUNTIME_ERROR;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

// `count` independent linkage problems (recordings) of dimension d in one call: their serial merge chains advance together
// (one launch = one round of every unfinished problem).  data[k]: n[k] x d row-major, dendrograms[k]: (n[k] - 1) x 4 — HOST
// pointers unless device_pointers != 0 (the two pointer ARRAYS are always host arrays).  statuses[k] (nullable) carries the
// per-problem status of the reference contract (n == 0 or 1 -> SUCCESS, nothing written); the return value is the first failure.
fa_status fa_ahc_linkage_batch(fa_ctx *ctx, int32_t count, const double *const *data, const size_t *n, size_t d, double *const *dendrograms,
                               int32_t mode, int32_t device_pointers, fa_ahc_stats *stats, int32_t *statuses) {
    if (!ctx || count < 0 || (count > 0 && (!data || !n || !dendrograms))) return FA_INVALID_ARGUMENT;
    if (count == 0) return FA_SUCCESS;
    try {
        fa::DeviceGuard guard(ctx->device);
        std::vector<fa_status> st(count, FA_SUCCESS);
        std::vector<const double *> d_in(count, nullptr);
        std::vector<double *> d_z(count, nullptr);
        std::vector<size_t> nn(count, 0);
        std::vector<fa::DevBuf> bufs(static_cast<size_t>(2) * count);
        for (int k = 0; k < count; ++k) {
            bool trivial;
            st[k] = linkage_checks(data[k], n[k], d, dendrograms[k], n[k] > 1 ? (n[k] - 1) * 4 : 0, &trivial);
            if (st[k] != FA_SUCCESS || trivial) continue;
            nn[k] = n[k];
            if (device_pointers) { d_in[k] = data[k]; d_z[k] = dendrograms[k]; continue; }
            if (bufs[2 * k].alloc(sizeof(double) * n[k] * d) != hipSuccess || bufs[2 * k + 1].alloc(sizeof(double) * 4 * (n[k] - 1)) != hipSuccess) {
                (void)hipGetLastError();
                st[k] = FA_ALLOCATION_FAILURE; nn[k] = 0;
                continue;
            }
            FA_HIP_TRY(ctx, hipMemcpyAsync(bufs[2 * k].p, data[k], sizeof(double) * n[k] * d, hipMemcpyHostToDevice, ctx->stream));
            d_in[k] = bufs[2 * k].as<double>(); d_z[k] = bufs[2 * k + 1].as<double>();
        }
        std::vector<fa_status> run(count, FA_SUCCESS);
        (void)fa::ahc_run_device_batch(ctx, count, d_in.data(), nn.data(), d, d_z.data(), mode, stats, run.data());
        fa_status first = FA_SUCCESS;
        for (int k = 0; k < count; ++k) {
            if (st[k] == FA_SUCCESS && nn[k] >= 2) st[k] = run[k];
            if (st[k] == FA_SUCCESS && nn[k] >= 2 && !device_pointers)
                FA_HIP_TRY(ctx, hipMemcpyAsync(dendrograms[k], d_z[k], sizeof(double) * 4 * (nn[k] - 1), hipMemcpyDeviceToHost, ctx->stream));
            if (statuses) statuses[k] = st[k];
            if (st[k] != FA_SUCCESS && first == FA_SUCCESS) first = st[k];
        }
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return first;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FA_RUNTIME_ERROR;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fastcluster_wrapper_status fastcluster_compute_centroid_linkage(const double *data, size_t pointCount, size_t dimension,
                                                                double *dendrogramOut, size_t dendrogramLength) {
    bool trivial;
    const fa_status pre = linkage_checks(data, pointCount, dimension, dendrogramOut, dendrogramLength, &trivial);
    if (pre != FA_SUCCESS || trivial) return static_cast<fastcluster_wrapper_status>(pre);
    try {
        // re-entrant from any thread: every call borrows one context of the default device set (pool.hip) for its duration, so
        // concurrent callers run on different GPUs (FLUIDAUDIO_HIP_DEVICES) and queue only when all of them are taken
        fa_pool *pool = nullptr;
        const fa_status ps = fa::default_pool(&pool);
        if (ps != FA_SUCCESS) return static_cast<fastcluster_wrapper_status>(ps == FA_ALLOCATION_FAILURE ? ps : FA_RUNTIME_ERROR);
        fa_ctx *ctx = nullptr;
        if (fa_pool_acquire(pool, &ctx) != FA_SUCCESS) return FASTCLUSTER_WRAPPER_RUNTIME_ERROR;
        struct Release { fa_pool *p; fa_ctx *c; ~Release() { fa_pool_release(p, c); } } release{pool, ctx};
        return static_cast<fastcluster_wrapper_status>(
            fa_ahc_linkage(ctx, data, pointCount, dimension, dendrogramOut, dendrogramLength, FA_AHC_MODE_AUTO, 0, nullptr));
    } catch (const std::bad_alloc &) {
        return FASTCLUSTER_WRAPPER_ALLOCATION_FAILURE;
    } catch (const std::exception &) {
        return FASTCLUSTER_WRAPPER_RUNTIME_ERROR;
    } catch (...) {
        return FASTCLUSTER_WRAPPER_UNKNOWN_ERROR;
    }
}

fa_status fa_ctx_reserve(fa_ctx *ctx, size_t n_max, size_t d, int32_t recordings) {
    if (!ctx || d == 0 || recordings < 1) return FA_INVALID_ARGUMENT;
    if (n_max < 2) return FA_SUCCESS;
    return fa::no_throw(ctx, "fa_ctx_reserve", [&]() -> fa_status {
        fa::DeviceGuard guard(ctx->device);
        FA_TRY(prob_check_shape(ctx, n_max, d));
        if (recordings == 1) {
            const size_t Np = (n_max + kBlk - 1) / kBlk * kBlk;
            fa::WsUse use(ctx);
            return fa::ws_acquire(ctx, make_layout(n_max, Np, d, Np / kBlk).total);
        }
        // The reservation follows the dispatch of a batch of `recordings` problems of n_max points (run_device_batch_impl): six or more long recordings
        // run as two uniform batches side by side, the second on a helper context with a workspace of its OWN — reserved here as well, so that the
        // first request pays no hipMalloc on either (until round 5 everything was reserved on the caller's context: the helper still allocated inside
        // the first request, and the two together held ~1.5 x the need).  The slot size is that of the round kernel the batch will run with.
        std::vector<size_t> n(static_cast<size_t>(recordings), n_max);
        const bool capped = ctx->ws_cap != static_cast<size_t>(-1);
        int groups = !capped && uniform_eligible(recordings, n.data(), FA_AHC_MODE_AUTO) ? uniform_groups(recordings, n.data()) : 1;
        for (int g = 1; g < groups; ++g) {
            fa_ctx *&h = ctx->helpers[g - 1];
            if (!h) {
                if (fa_ctx_create(ctx->device, nullptr, &h) != FA_SUCCESS) { h = nullptr; groups = g; break; }
                h->ws_limit = ctx->ws_limit;
                h->ws_cap = ctx->ws_cap;
            }
        }
        for (int g = 0; g < groups; ++g) {
            const int m = static_cast<int>(static_cast<long long>(recordings) * (g + 1) / groups - static_cast<long long>(recordings) * g / groups);
            fa_ctx *c = g == 0 ? ctx : ctx->helpers[g - 1];
            fa::WsUse use(c);
            const fa_status st = fa::ws_acquire(c, uniform_stride(m, n_max, d) * static_cast<size_t>(m));
            if (st != FA_SUCCESS) { if (c != ctx) ctx->last_error = c->last_error; return st; }
        }
        return FA_SUCCESS;
    });
}

fa_status fa_ahc_cut(const double *z, size_t n, double threshold, int32_t *labels) {
    // AHCClustering.swift:112-121 (clamp), :124-197 (top-down cut), :200-210 (relabel by first appearance)
    if (n == 0) return FA_SUCCESS;
    if (!labels || (n > 1 && !z)) return FA_INVALID_ARGUMENT;
    if (n == 1) { labels[0] = 0; return FA_SUCCESS; }
    try {
        double thr = threshold;
        if (thr != thr) thr = 0.0;
        thr = std::max(0.0, std::min(2.0, thr));
        const size_t total = 2 * n - 1;
        std::vector<int64_t> left(total, -1), right(total, -1), assign(n, -1);
        std::vector<double> height(total, 0.0);
        std::vector<char> merged(total, 0);
        for (size_t r = 0; r + 1 < n; ++r) {
            // The reference only ever cuts what its own wrapper wrote (AHCClustering.swift:40-58); a C caller can hand over anything.  The
            // children of row r must be two different nodes that exist when it is formed (leaves, or rows < r) and were not merged before:
            // anything else is an out-of-bounds read or an endless walk below.
            const double a = z[4 * r], b = z[4 * r + 1], limit = static_cast<double>(n + r);
            if (!(a >= 0.0 && a < limit && b >= 0.0 && b < limit) || a != std::floor(a) || b != std::floor(b) || a == b) return FA_INVALID_ARGUMENT;
            const size_t ia = static_cast<size_t>(a), ib = static_cast<size_t>(b);
            if (merged[ia] || merged[ib]) return FA_INVALID_ARGUMENT;
            merged[ia] = merged[ib] = 1;
            left[n + r] = static_cast<int64_t>(ia);
            right[n + r] = static_cast<int64_t>(ib);
            height[n + r] = z[4 * r + 2];
        }
        std::vector<int64_t> stack{static_cast<int64_t>(total - 1)}, queue;
        int64_t next = 0;
        while (!stack.empty()) {
            const int64_t node = stack.back();
            stack.pop_back();
            if (node < 0) continue;
            if (node < static_cast<int64_t>(n)) { if (assign[node] == -1) assign[node] = next++; continue; }
            if (height[node] <= thr) {
                const int64_t label = next++;
                queue.assign(1, node);
                while (!queue.empty()) {
                    const int64_t cur = queue.back();
                    queue.pop_back();
                    if (cur < static_cast<int64_t>(n)) assign[cur] = label;
                    else { if (left[cur] >= 0) queue.push_back(left[cur]); if (right[cur] >= 0) queue.push_back(right[cur]); }
                }
            } else {
                if (left[node] >= 0) stack.push_back(left[node]);
                if (right[node] >= 0) stack.push_back(right[node]);  // popped first => right subtree visited first
            }
        }
        for (size_t i = 0; i < n; ++i) if (assign[i] == -1) assign[i] = next++;
        std::vector<int32_t> remap(static_cast<size_t>(next), -1);
        int32_t nid = 0;
        for (size_t i = 0; i < n; ++i) {
            if (remap[assign[i]] < 0) remap[assign[i]] = nid++;
            labels[i] = remap[assign[i]];
        }
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fa_status fa_ahc_cluster(fa_ctx *ctx, const double *x, size_t n, size_t d, double threshold, int32_t mode, int32_t *labels,
                         fa_ahc_stats *stats) {
    // AHCClustering.swift:20-67
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (!labels) return FA_INVALID_ARGUMENT;
    if (d == 0) { for (size_t i = 0; i < n; ++i) labels[i] = 0; return FA_SUCCESS; }
    if (!x) return FA_INVALID_ARGUMENT;
    if (n == 1) { labels[0] = 0; return FA_SUCCESS; }
    try {
        std::vector<double> norm(n * d), z((n - 1) * 4, 0.0);
        for (size_t i = 0; i < n; ++i) {  // normalizeFeatures (:70-105)
#pragma clang fp contract(off)
            const double *row = x + i * d;
            double ss = 0.0;
            for (size_t k = 0; k < d; ++k) ss += row[k] * row[k];
            const double scale = ss > 0 ? 1.0 / std::sqrt(ss) : 0.0;
            for (size_t k = 0; k < d; ++k) norm[i * d + k] = row[k] * scale;
        }
        const fa_status st = fa_ahc_linkage(ctx, norm.data(), n, d, z.data(), z.size(), mode, 0, stats);
        if (st != FA_SUCCESS) {
            for (size_t i = 0; i < n; ++i) labels[i] = static_cast<int32_t>(i);  // degrade, don't crash (:52-55)
            return st;
        }
        return fa_ahc_cut(z.data(), n, threshold, labels);
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Row minima of a SLAB of the pairwise distance matrix: for rows [row0, row1) the nearest other point among all n (the
// reference's distance: sequential fp64 sum of squared differences, FastClusterWrapper.cpp:45-52; lowest index on ties).
// This is the start-up of the linkage (fastcluster_internal.hpp:1653-1678 builds the same nearest-neighbour table) in a form
// that shards by rows across GPUs (all-gather X, per-rank slab, gather (min, idx)); fluidaudio_amd/sharding.py
// drives it.  64 x 64 output tile per workgroup step, 4 x 4 per thread, operands staged k-major in LDS.
namespace {

constexpr int kSlabT = 64, kSlabK = 16;

__global__ __launch_bounds__(256) void slab_row_minima_kernel(const double *__restrict__ x, int n, int d, int row0, int row1, double *__restrict__ out_min,
                                                              int32_t *__restrict__ out_arg) {
    __shared__ double sa[kSlabK][kSlabT + 1], sb[kSlabK][kSlabT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // tx: column quad, ty: row quad
    const int i0 = row0 + blockIdx.x * kSlabT;
    double best[4];
    int arg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { best[r] = __longlong_as_double(0x7ff0000000000000LL); arg[r] = -1; }
    for (int j0 = 0; j0 < n; j0 += kSlabT) {
        double acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
        for (int k0 = 0; k0 < d; k0 += kSlabK) {
            for (int e = tid; e < kSlabT * kSlabK; e += 256) {   // 64 rows x 16 dims of both operands
                const int rr = e / kSlabK, kk = e % kSlabK;
                const int gi = i0 + rr, gj = j0 + rr, gk = k0 + kk;
                sa[kk][rr] = gi < row1 && gk < d ? x[static_cast<size_t>(gi) * d + gk] : 0.0;
                sb[kk][rr] = gj < n && gk < d ? x[static_cast<size_t>(gj) * d + gk] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < kSlabK; ++kk) {   // ascending k, one rounding per operation: the reference's sum
                double a[4], b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = sa[kk][4 * ty + r]; b[r] = sb[kk][4 * tx + r]; }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const double df = a[r] - b[c]; acc[r][c] = acc[r][c] + df * df; }
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gi = i0 + 4 * ty + r, gj = j0 + 4 * tx + c;
                if (gi < row1 && gj < n && gj != gi && (acc[r][c] < best[r] || (acc[r][c] == best[r] && gj < arg[r]))) { best[r] = acc[r][c]; arg[r] = gj; }
            }
    }
    // the 16 threads of a row quad (tx = 0..15, consecutive lanes) combine: lowest value, then lowest index
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = best[r];
        int a = arg[r];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, 16);
            const int oa = __shfl_xor(a, off, 16);
            if (oa >= 0 && (a < 0 || ov < v || (ov == v && oa < a))) { v = ov; a = oa; }
        }
        const int gi = i0 + 4 * ty + r;
        if (tx == 0 && gi < row1) { out_min[gi - row0] = v; out_arg[gi - row0] = a; }
    }
}

}  // namespace

extern "C" fa_status fa_ahc_row_minima(fa_ctx *ctx, const double *x, size_t n, size_t d, size_t row0, size_t row1, double *mins, int32_t *args,
                                       int32_t device_pointers) {
    if (!ctx || !x || !mins || !args) return FA_INVALID_ARGUMENT;
    if (row0 > row1 || row1 > n || d == 0 || n > static_cast<size_t>(INT32_MAX) || d > static_cast<size_t>(INT32_MAX)) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "row minima: bad range");
    if (row0 == row1) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    const size_t rows = row1 - row0;
    fa::DevBuf bx, bm, ba;
    const double *d_x = x;
    double *d_m = mins;
    int32_t *d_a = args;
    if (!device_pointers) {
        if (bx.alloc(sizeof(double) * n * d) != hipSuccess || bm.alloc(sizeof(double) * rows) != hipSuccess || ba.alloc(sizeof(int32_t) * rows) != hipSuccess) {
            (void)hipGetLastError();
            return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "row minima: device allocation failed");
        }
        FA_HIP_TRY(ctx, hipMemcpyAsync(bx.p, x, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream));
        d_x = bx.as<double>(); d_m = bm.as<double>(); d_a = ba.as<int32_t>();
    }
    hipLaunchKernelGGL(slab_row_minima_kernel, dim3(static_cast<unsigned>((rows + kSlabT - 1) / kSlabT)), dim3(256), 0, ctx->stream, d_x, static_cast<int>(n),
                       static_cast<int>(d), static_cast<int>(row0), static_cast<int>(row1), d_m, d_a);
    FA_HIP_TRY(ctx, hipGetLastError());
    if (!device_pointers) {
        FA_HIP_TRY(ctx, hipMemcpyAsync(mins, d_m, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipMemcpyAsync(args, d_a, sizeof(int32_t) * rows, hipMemcpyDeviceToHost, ctx->stream));
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FA_SUCCESS;
}

