// vbx.hip — VBx variational-Bayes refinement of AHC labels (fp64) on gfx950.
//
// Replaces VBxClustering.refine / runVBx
// (reference: Sources/FluidAudio/Diarizer/Offline/Clustering/VBxClustering.swift:41-165,167-664).
// The reference's two DGEMMs per iteration (gamma^T rho: S x D <- T; rho alpha^T: T x S) and its
// row soft-max are small and skinny (T ~ 4e4, D = 128, S = #AHC clusters); they are evaluated
// here with deterministic reductions (fixed split of T, partials summed in a fixed order) so
// that repeated runs are bit-identical.  fp64 throughout, like the reference (cblas_d*, vvexp).
#include <algorithm>
#include <cmath>
#include <vector>

#include "fa_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSplit = 64;  // fixed split of the frame axis for the S x D contraction

struct VbxWs {
    const double *X;   // [T][D] input features (rho of the reference)
    const double *phi; // [D] (already clamped to >= 1e-12)
    double *rho;       // [T][D] = X * sqrt(phi)
    double *G;         // [T]
    double *gamma;     // [T][S]
    double *pi;        // [S]
    double *logpi;     // [S]
    double *part;      // [kSplit][S][D+1]  (column D carries sum_t gamma)
    double *alpha;     // [S][D]
    double *invL;      // [S][D]
    double *phiT;      // [S]
    double *llrow;     // [T]
    double *scal;      // [8]: 0 elbo, 1 ll
    int64_t T;
    int32_t D, S;
    double Fa, Fb;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}

// rho = X * sqrt(phi) (:242-265); G[t] = -0.5 (||x_t||^2 + D ln 2 pi) (:267-282).  One wave per frame.
__global__ void vbx_prepare(VbxWs w) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (t >= w.T) return;
    const int lane = threadIdx.x & 63;
    double ss = 0.0;
    for (int d = lane; d < w.D; d += 64) {
        const double x = w.X[t * w.D + d];
        w.rho[t * w.D + d] = x * sqrt(w.phi[d]);
        ss += x * x;
    }
    ss = wave_sum(ss);
    if (lane == 0) w.G[t] = -0.5 * (ss + static_cast<double>(w.D) * log(2.0 * M_PI));
}

// one-hot from labels -> softmax(7 * onehot) -> renormalise (:102-107, :190-237). One wave per frame.
__global__ void vbx_init_gamma(VbxWs w, const int32_t *labels, double smoothing) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (t >= w.T) return;
    const int lane = threadIdx.x & 63;
    int sp = labels[t];
    sp = sp > w.S - 1 ? w.S - 1 : sp;
    sp = sp < 0 ? 0 : sp;
    double *g = w.gamma + t * w.S;
    // row max of smoothing*onehot is `smoothing` (or 0 when smoothing < 0 is not used by the reference)
    const double mx = smoothing > 0.0 ? smoothing : 0.0;
    double sum = 0.0;
    for (int s = lane; s < w.S; s += 64) sum += exp((s == sp ? smoothing : 0.0) - mx);
    sum = wave_sum(sum);
    const double inv = 1.0 / sum;
    double sum2 = 0.0;
    for (int s = lane; s < w.S; s += 64) { const double v = exp((s == sp ? smoothing : 0.0) - mx) * inv; g[s] = v; sum2 += v; }
    sum2 = wave_sum(sum2);
    const double inv2 = 1.0 / sum2;
    for (int s = lane; s < w.S; s += 64) g[s] *= inv2;
}

__global__ void vbx_fill(double *p, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// part[z][s][0..D] = sum over the z-th slice of frames of gamma[t][s] * (rho[t][:], 1)   (:312-325, :342-357)
// grid: (ceil((D+1)/64), ceil(S/4), kSplit), block 256 = 64 columns x 4 speakers.
__global__ __launch_bounds__(kThreads) void vbx_gt_rho(VbxWs w, int first_col_tile) {
    const int col = (blockIdx.x + first_col_tile) * 64 + (threadIdx.x & 63);
    const int s = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.z;
    const int64_t per = (w.T + kSplit - 1) / kSplit;
    const int64_t t0 = z * per, t1 = t0 + per < w.T ? t0 + per : w.T;
    if (s >= w.S || col > w.D) return;
    double acc = 0.0;
    if (col < w.D) for (int64_t t = t0; t < t1; ++t) acc += w.gamma[t * w.S + s] * w.rho[t * w.D + col];
    else for (int64_t t = t0; t < t1; ++t) acc += w.gamma[t * w.S + s];
    w.part[(static_cast<int64_t>(z) * w.S + s) * (w.D + 1) + col] = acc;
}

// Per speaker: N_s, invL, alpha, phi term (:330-337, :370-387, :402-432).  One workgroup per speaker.
// mode 0: E-step quantities; mode 1: only pi[s] = sum_t gamma (used after the soft-max, :586-603).
__global__ __launch_bounds__(kThreads) void vbx_speaker(VbxWs w, int mode) {
    __shared__ double red[kThreads / 64];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int D = w.D;
    if (mode == 1) {
        if (tid == 0) {
            double ns = 0.0;
            for (int z = 0; z < kSplit; ++z) ns += w.part[(static_cast<int64_t>(z) * w.S + s) * (D + 1) + D];
            w.pi[s] = ns;
        }
        return;
    }
    double ns = 0.0;
    for (int z = 0; z < kSplit; ++z) ns += w.part[(static_cast<int64_t>(z) * w.S + s) * (D + 1) + D];
    const double weight = (w.Fa / w.Fb) * ns;
    double acc = 0.0;
    for (int d = tid; d < D; d += kThreads) {
        double tmp = 0.0;
        for (int z = 0; z < kSplit; ++z) tmp += w.part[(static_cast<int64_t>(z) * w.S + s) * (D + 1) + d];
        const double den = 1.0 + weight * w.phi[d];
        const double il = 1.0 / (den > 1e-12 ? den : 1e-12);
        const double al = (tmp * il) * (w.Fa / w.Fb);
        w.invL[static_cast<int64_t>(s) * D + d] = il;
        w.alpha[static_cast<int64_t>(s) * D + d] = al;
        acc += (al * al + il) * w.phi[d];
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) { double v = 0.0; for (int i = 0; i < kThreads / 64; ++i) v += red[i]; w.phiT[s] = v; }
}

__global__ void vbx_logpi(VbxWs w) {  // log(max(pi, 1e-8)) (:498-514)
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < w.S) { const double p = w.pi[s]; w.logpi[s] = log(p >= 1e-8 ? p : 1e-8); }
}

// E-step for one frame per wave: logP[s] = Fa (rho_t . alpha_s - phiT_s/2 + G_t) (:441-492),
// gamma = softmax(logP + log pi), llrow = logsumexp (:516-572).  S is processed in chunks of 64.
__global__ __launch_bounds__(kThreads) void vbx_estep(VbxWs w) {
    extern __shared__ double sm[];  // [4 waves][D] rho rows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    if (t >= w.T) return;
    double *rt = sm + wave * w.D;
    for (int d = lane; d < w.D; d += 64) rt[d] = w.rho[t * w.D + d];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double *g = w.gamma + t * w.S;
    const double gt = w.G[t];
    double mx = -1.7976931348623157e308;
    for (int s = lane; s < w.S; s += 64) {
        const double *al = w.alpha + static_cast<int64_t>(s) * w.D;
        double dot = 0.0;
        for (int d = 0; d < w.D; ++d) dot += rt[d] * al[d];
        const double lp = ((dot + w.phiT[s] * -0.5) + gt) * w.Fa + w.logpi[s];
        g[s] = lp;  // staged in place; overwritten below
        mx = lp > mx ? lp : mx;
    }
    mx = wave_max(mx);
    double sum = 0.0;
    for (int s = lane; s < w.S; s += 64) { const double e = exp(g[s] - mx); g[s] = e; sum += e; }
    sum = wave_sum(sum);
    if (sum <= 0.0 || !isfinite(sum)) {
        for (int s = lane; s < w.S; s += 64) g[s] = 1.0 / static_cast<double>(w.S);
        if (lane == 0) w.llrow[t] = mx;
    } else {
        const double inv = 1.0 / sum;
        for (int s = lane; s < w.S; s += 64) g[s] *= inv;
        if (lane == 0) w.llrow[t] = mx + log(sum);
    }
}

// Scalars of one iteration: normalise pi (:605-621), log-likelihood (sum of llrow in frame order) and
// ELBO = ll + Fb/2 * sum(log invL - invL - alpha^2 + 1) (:623-647).  One workgroup, fixed order.
__global__ __launch_bounds__(kThreads) void vbx_scalars(VbxWs w) {
    __shared__ double red[kThreads];
    const int tid = threadIdx.x;
    double a = 0.0;
    for (int64_t t = tid; t < w.T; t += kThreads) a += w.llrow[t];
    red[tid] = a;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    const double ll = red[0];
    __syncthreads();
    double b = 0.0;
    const int64_t n = static_cast<int64_t>(w.S) * w.D;
    for (int64_t i = tid; i < n; i += kThreads) { const double il = w.invL[i], al = w.alpha[i]; b += log(il) - il - al * al + 1.0; }
    red[tid] = b;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    if (tid == 0) {
        w.scal[0] = ll + w.Fb * 0.5 * red[0];
        w.scal[1] = ll;
        double ps = 0.0;
        for (int s = 0; s < w.S; ++s) ps += w.pi[s];
        if (ps > 0.0 && isfinite(ps)) { const double inv = 1.0 / ps; for (int s = 0; s < w.S; ++s) w.pi[s] *= inv; }
        else for (int s = 0; s < w.S; ++s) w.pi[s] = 1.0 / static_cast<double>(w.S);
    }
}

__global__ void vbx_hard(VbxWs w, int32_t *hard) {  // first maximum (:144-146)
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= w.T) return;
    const double *g = w.gamma + t * w.S;
    int b = 0;
    for (int s = 1; s < w.S; ++s) if (g[b] < g[s]) b = s;
    hard[t] = b;
}

}  // namespace

extern "C" {

int32_t fa_vbx_speaker_count(const int32_t *initial, int64_t T) {
    if (!initial || T <= 0) return 0;
    try {
        std::vector<int32_t> tmp(initial, initial + T);
        std::sort(tmp.begin(), tmp.end());
        const int64_t n = std::unique(tmp.begin(), tmp.end()) - tmp.begin();
        return static_cast<int32_t>(n < 1 ? 1 : n);  // max(1, Set(initialClusters).count) (:78)
    } catch (...) {
        return 0;
    }
}

fa_status fa_vbx_refine(fa_ctx *ctx, const double *rho, int64_t T, int32_t D, const int32_t *initial, const double *phi,
                        double Fa, double Fb, int32_t max_iter, double epsilon, double *gamma, double *pi, int32_t *hard,
                        double *elbos, int32_t *n_iters, int32_t *n_speakers) {
    if (!ctx || !n_iters || !n_speakers) return FA_INVALID_ARGUMENT;
    *n_iters = 0;
    *n_speakers = 0;
    if (T <= 0 || D <= 0) return FA_SUCCESS;  // empty VBxOutput (:45-67)
    if (!rho || !initial || !phi || !gamma || !pi || !hard || (max_iter > 0 && !elbos)) return FA_INVALID_ARGUMENT;
    const int32_t S = fa_vbx_speaker_count(initial, T);
    if (S < 1) return FA_ALLOCATION_FAILURE;
    *n_speakers = S;
    fa::DeviceGuard guard(ctx->device);
    try {
        const size_t TD = static_cast<size_t>(T) * D, TS = static_cast<size_t>(T) * S;
        fa::DevBuf bX, blab;
        if (bX.alloc(8 * TD) != hipSuccess || blab.alloc(4 * T) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx: device allocation failed"); }
        hipStream_t st = ctx->stream;
        FA_HIP_TRY(ctx, hipMemcpyAsync(bX.p, rho, 8 * TD, hipMemcpyHostToDevice, st));
        FA_HIP_TRY(ctx, hipMemcpyAsync(blab.p, initial, 4 * T, hipMemcpyHostToDevice, st));
        fa::VbxDevice dev;
        FA_TRY(fa::vbx_run_dev(ctx, bX.as<double>(), T, D, blab.as<int32_t>(), S, phi, Fa, Fb, max_iter, epsilon, elbos, n_iters, dev));
        FA_HIP_TRY(ctx, hipMemcpyAsync(gamma, dev.gamma.p, 8 * TS, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipMemcpyAsync(pi, dev.pi.p, 8 * S, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipMemcpyAsync(hard, dev.hard.p, 4 * T, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx: host allocation failed");
    } catch (...) {
        return fa::set_error(ctx, FA_UNKNOWN_ERROR, "vbx: unexpected failure");
    }
}

}  // extern "C"

// The EM loop on device-resident inputs (d_X: [T][D] rho features, d_labels: [T] AHC labels with S distinct values); gamma, pi and
// the hard assignment stay on the device in `out`.  The ELBO of every iteration crosses to the host (8 bytes) for the
// convergence test of the reference (:653-659).
fa_status fa::vbx_run_dev(fa_ctx *ctx, const double *d_X, int64_t T, int32_t D, const int32_t *d_labels, int32_t S, const double *phi_host,
                          double Fa, double Fb, int32_t max_iter, double epsilon, double *elbos, int32_t *n_iters, fa::VbxDevice &o) {
    if (n_iters) *n_iters = 0;
    std::vector<double> phic(D);
    for (int d = 0; d < D; ++d) phic[d] = phi_host[d] > 1e-12 ? phi_host[d] : 1e-12;  // :241
    const size_t TD = static_cast<size_t>(T) * D, TS = static_cast<size_t>(T) * S, SD = static_cast<size_t>(S) * D;
    hipError_t e = hipSuccess;
    auto A = [&](fa::DevBuf &b, size_t bytes) { if (e == hipSuccess) e = b.alloc(bytes); };
    A(o.phi, 8 * D); A(o.rho, 8 * TD); A(o.G, 8 * T); A(o.gamma, 8 * TS); A(o.pi, 8 * S); A(o.logpi, 8 * S);
    A(o.part, 8 * static_cast<size_t>(kSplit) * S * (D + 1)); A(o.alpha, 8 * SD); A(o.invL, 8 * SD); A(o.phiT, 8 * S);
    A(o.ll, 8 * T); A(o.scal, 64); A(o.hard, 4 * T);
    if (e != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx: device allocation failed"); }
    o.T = T; o.D = D; o.S = S;
    hipStream_t st = ctx->stream;
    FA_HIP_TRY(ctx, hipMemcpyAsync(o.phi.p, phic.data(), 8 * D, hipMemcpyHostToDevice, st));
    VbxWs w{};
    w.X = d_X; w.phi = o.phi.as<double>(); w.rho = o.rho.as<double>(); w.G = o.G.as<double>();
    w.gamma = o.gamma.as<double>(); w.pi = o.pi.as<double>(); w.logpi = o.logpi.as<double>(); w.part = o.part.as<double>();
    w.alpha = o.alpha.as<double>(); w.invL = o.invL.as<double>(); w.phiT = o.phiT.as<double>(); w.llrow = o.ll.as<double>();
    w.scal = o.scal.as<double>(); w.T = T; w.D = D; w.S = S; w.Fa = Fa; w.Fb = Fb;
    const int wave_blocks = static_cast<int>((T + 3) / 4);
    hipLaunchKernelGGL(vbx_prepare, dim3(wave_blocks), dim3(kThreads), 0, st, w);
    hipLaunchKernelGGL(vbx_init_gamma, dim3(wave_blocks), dim3(kThreads), 0, st, w, d_labels, 7.0);
    hipLaunchKernelGGL(vbx_fill, dim3((S + 255) / 256), dim3(256), 0, st, w.pi, S, 1.0 / static_cast<double>(S));  // :239
    FA_HIP_TRY(ctx, hipGetLastError());
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));   // phic (a host temporary) has been consumed
    const dim3 ggrid((D + 1 + 63) / 64, (S + 3) / 4, kSplit);
    const size_t estep_lds = sizeof(double) * 4 * static_cast<size_t>(D);
    if (estep_lds > 64 * 1024) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "vbx: feature dimension too large");
    double prev = -1.7976931348623157e308;
    int iters = 0;
    for (int it = 0; it < max_iter; ++it) {
        iters = it + 1;
        hipLaunchKernelGGL(vbx_gt_rho, ggrid, dim3(kThreads), 0, st, w, 0);
        hipLaunchKernelGGL(vbx_speaker, dim3(S), dim3(kThreads), 0, st, w, 0);
        hipLaunchKernelGGL(vbx_logpi, dim3((S + 255) / 256), dim3(256), 0, st, w);
        hipLaunchKernelGGL(vbx_estep, dim3(wave_blocks), dim3(kThreads), estep_lds, st, w);
        // only the column tile that holds column D of the partials: sum_t gamma of the NEW gamma (pi, :586-603)
        hipLaunchKernelGGL(vbx_gt_rho, dim3(1, ggrid.y, kSplit), dim3(kThreads), 0, st, w, D / 64);
        hipLaunchKernelGGL(vbx_speaker, dim3(S), dim3(kThreads), 0, st, w, 1);
        hipLaunchKernelGGL(vbx_scalars, dim3(1), dim3(kThreads), 0, st, w);
        FA_HIP_TRY(ctx, hipGetLastError());
        double elbo = 0.0;
        FA_HIP_TRY(ctx, hipMemcpyAsync(&elbo, w.scal, sizeof(double), hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (elbos) elbos[it] = elbo;
        if (it > 0 && std::fabs(elbo - prev) < epsilon) { prev = elbo; break; }  // :653-659
        prev = elbo;
    }
    hipLaunchKernelGGL(vbx_hard, dim3(static_cast<int>((T + 255) / 256)), dim3(256), 0, st, w, o.hard.as<int32_t>());
    FA_HIP_TRY(ctx, hipGetLastError());
    if (n_iters) *n_iters = iters;
    return FA_SUCCESS;
}
