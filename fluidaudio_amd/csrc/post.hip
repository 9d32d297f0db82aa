// post.hip — what OfflineDiarizerManager.cluster does with the VBx posteriors (gfx950, fp64).
//
// Replaces computeCentroids (reference: Sources/FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:613-691:
// gamma-weighted mean of the 256-d embeddings for every speaker with pi > 1e-7) and assignEmbeddings (:789-822: cosine
// argmax of every embedding against the centroids, first maximum wins; normalisation :824-859).
// Both are tiny next to AHC/VBx (N ~ 4e4, K <= ~100, d = 256), so the kernels keep the reference's summation ORDER
// (sequential in t for the centroid sums, sequential in the dimension for norms and dot products) and are therefore
// bit-identical to the CPU restatement, not just close: one thread owns one output element and walks the reduction
// axis; the loads of different iterations are independent, so they pipeline.
#include <cmath>
#include <vector>

#include "fa_common.h"

namespace {

constexpr int kThreads = 256;

// centroid[K][d]: thread (speaker slot c, dimension k).  num[k] += w * emb[t][k] with one rounding per operation (cblas_daxpy :655-672)
__global__ void centroid_kernel(const double *__restrict__ emb, const double *__restrict__ gamma, const int32_t *__restrict__ spk,
                                double *__restrict__ cent, int64_t n, int d, int S, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= d || c >= K) return;
    const int s = spk[c];
    double num = 0.0, den = 0.0;
    for (int64_t t = 0; t < n; ++t) {
        const double w = gamma[t * S + s];
        if (!(w > 0)) continue;
        den = __dadd_rn(den, w);
        num = __dadd_rn(num, __dmul_rn(w, emb[t * d + k]));
    }
    cent[static_cast<int64_t>(c) * d + k] = den > 0 ? __ddiv_rn(num, den) : 0.0;
}

// unit-normalised copy (normalize :824-859: a vector with sum of squares <= 0 is returned unchanged)
__global__ void normalize_rows(const double *__restrict__ v, double *__restrict__ out, int64_t rows, int d) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double *x = v + r * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss = __dadd_rn(ss, __dmul_rn(x[k], x[k]));
    const double scale = ss <= 0 ? 1.0 : __ddiv_rn(1.0, __dsqrt_rn(ss));
    for (int k = 0; k < d; ++k) out[r * d + k] = ss <= 0 ? x[k] : __dmul_rn(x[k], scale);
}

// argmax_k <e_i, c_k> over normalised vectors, first maximum (strict '>', :806-816)
__global__ void assign_kernel(const double *__restrict__ emb, const double *__restrict__ cn, int32_t *__restrict__ out, int64_t n,
                              int d, int K) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *x = emb + i * d;
    double ss = 0.0;
    for (int k = 0; k < d; ++k) ss = __dadd_rn(ss, __dmul_rn(x[k], x[k]));
    const double scale = ss <= 0 ? 1.0 : __ddiv_rn(1.0, __dsqrt_rn(ss));
    const bool keep = ss <= 0;
    double best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < K; ++c) {
        const double *cv = cn + static_cast<int64_t>(c) * d;
        double dot = 0.0;
        for (int k = 0; k < d; ++k) {
            const double e = keep ? x[k] : __dmul_rn(x[k], scale);
            dot = __dadd_rn(dot, __dmul_rn(e, cv[k]));
        }
        if (dot > best) { best = dot; bi = c; }
    }
    out[i] = bi;
}

}  // namespace

extern "C" {

fa_status fa_vbx_weighted_centroids(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *gamma, const double *pi,
                                    int32_t S, double *centroids, int32_t *map, int32_t *n_centroids) {
    if (!ctx || !n_centroids || !map || (S > 0 && !pi)) return FA_INVALID_ARGUMENT;
    *n_centroids = 0;
    if (n < 0 || d < 1 || S < 0) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "centroids: bad shape");
    try {
        std::vector<int32_t> spk;
        for (int s = 0; s < S; ++s) {  // speakers kept: pi > 1e-7 (:630-640)
            map[s] = -1;
            if (pi[s] > 1e-7) { map[s] = static_cast<int32_t>(spk.size()); spk.push_back(s); }
        }
        const int K = static_cast<int>(spk.size());
        *n_centroids = K;
        if (K == 0) return FA_SUCCESS;
        if (!centroids || (n > 0 && (!emb || !gamma))) return FA_INVALID_ARGUMENT;
        fa::DeviceGuard guard(ctx->device);
        fa::DevBuf d_emb, d_gamma, d_spk, d_cent;
        hipError_t e;
        do {
            if ((e = d_emb.alloc(sizeof(double) * n * d)) != hipSuccess) break;
            if ((e = d_gamma.alloc(sizeof(double) * n * S)) != hipSuccess) break;
            if ((e = d_spk.alloc(sizeof(int32_t) * K)) != hipSuccess) break;
            if ((e = d_cent.alloc(sizeof(double) * K * d)) != hipSuccess) break;
            if (n > 0 && (e = hipMemcpyAsync(d_emb.p, emb, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            if (n > 0 && (e = hipMemcpyAsync(d_gamma.p, gamma, sizeof(double) * n * S, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            if ((e = hipMemcpyAsync(d_spk.p, spk.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
            hipLaunchKernelGGL(centroid_kernel, dim3((d + 63) / 64, K), dim3(64), 0, ctx->stream, d_emb.as<double>(), d_gamma.as<double>(),
                               d_spk.as<int32_t>(), d_cent.as<double>(), n, d, S, K);
            if ((e = hipGetLastError()) != hipSuccess) break;
            if ((e = hipMemcpyAsync(centroids, d_cent.p, sizeof(double) * K * d, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
            e = hipStreamSynchronize(ctx->stream);
        } while (0);
        return fa::hip_status(ctx, e, "fa_vbx_weighted_centroids");
    } catch (const std::bad_alloc &) {
        return FA_ALLOCATION_FAILURE;
    } catch (...) {
        return FA_UNKNOWN_ERROR;
    }
}

fa_status fa_assign_cosine(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *centroids, int32_t K, int32_t *out) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n == 0) return FA_SUCCESS;
    if (n < 0 || d < 1 || K < 0 || !out || !emb) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "assign: bad arguments");
    if (K == 0) { for (int64_t i = 0; i < n; ++i) out[i] = 0; return FA_SUCCESS; }  // guard (:795-797)
    if (!centroids) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_emb, d_c, d_cn, d_out;
    hipError_t e;
    do {
        if ((e = d_emb.alloc(sizeof(double) * n * d)) != hipSuccess) break;
        if ((e = d_c.alloc(sizeof(double) * K * d)) != hipSuccess) break;
        if ((e = d_cn.alloc(sizeof(double) * K * d)) != hipSuccess) break;
        if ((e = d_out.alloc(sizeof(int32_t) * n)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_emb.p, emb, sizeof(double) * n * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_c.p, centroids, sizeof(double) * K * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        hipLaunchKernelGGL(normalize_rows, dim3((K + 63) / 64), dim3(64), 0, ctx->stream, d_c.as<double>(), d_cn.as<double>(), static_cast<int64_t>(K), d);
        hipLaunchKernelGGL(assign_kernel, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                           d_emb.as<double>(), d_cn.as<double>(), d_out.as<int32_t>(), n, d, K);
        if ((e = hipGetLastError()) != hipSuccess) break;
        if ((e = hipMemcpyAsync(out, d_out.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    return fa::hip_status(ctx, e, "fa_assign_cosine");
}

}  // extern "C"
