// vbx.hip — VBx variational-Bayes refinement of AHC labels (fp64) on gfx950.
//
// Replaces VBxClustering.refine / runVBx
// (reference: Sources/FluidAudio/Diarizer/Offline/Clustering/VBxClustering.swift:41-165,167-664).
// The reference's two DGEMMs per iteration (gamma^T rho: S x D <- T; rho alpha^T: T x S) and its
// row soft-max are small and skinny (T ~ 4e4, D = 128, S = #AHC clusters); they are evaluated
// here with deterministic reductions (fixed split of T, partials summed in a fixed order) so
// that repeated runs are bit-identical.  fp64 throughout, like the reference (cblas_d*, vvexp).
//
// Everything that crosses frames goes through ONE record per slice of the frame axis (kSplit = 64 slices): the slice's
// sum_t gamma[t][s] (rho[t][:], 1) and its sum of the per-frame log-likelihoods.  An iteration reads the 64 records in slice
// order (speaker statistics, pi, ELBO) and writes the 64 records of the new posteriors — so the same kernels run the whole
// problem on one device (vbx_run_dev) or a contiguous range of slices per device with one all-gather of the records per
// iteration in between (fa_vbx_shard_*: SURVEY §8(e) row 4, the iteration loop of VBxClustering.swift:301-661 sharded over
// T).  Both give the same bits: a shard computes exactly the records the single device computes for those slices.
#include <algorithm>
#include <cmath>
#include <cstdlib>
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
    const double *rec_in;  // [kSplit][stride] complete slice records: [S][D+1] (column D carries sum_t gamma), then the slice's sum of llrow
    double *rec_out;       // [z_n][stride] records of the slices z_lo .. z_lo + z_n - 1 that this device owns
    double *alpha;     // [S][D]
    double *invL;      // [S][D]
    double *phiT;      // [S]
    double *llrow;     // [T]
    double *scal;      // [8]: 0 elbo, 1 ll
    int64_t T;         // frames held here: the global frames t0g .. t0g + T - 1
    int64_t Tg, t0g;   // frames of the whole problem; global index of local frame 0
    int64_t stride;    // doubles per slice record = S (D + 1) + 1
    int32_t D, S;
    int32_t z_lo, z_n; // slices owned
    double Fa, Fb;
    int32_t tiled;     // host side: the tiled products serve S >= kVbxTiledMinS (FA_VBX_NO_TILED, read ONCE per refinement when the workspace is set up)
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

// record[z][s][0..D] = sum over the z-th slice of frames of gamma[t][s] * (rho[t][:], 1)   (:312-325, :342-357)
// grid: (ceil((D+1)/64), ceil(S/4), slices owned), block 256 = 64 columns x 4 speakers.
__global__ __launch_bounds__(kThreads) void vbx_gt_rho(VbxWs w) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int s = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = w.z_lo + blockIdx.z;
    const int64_t per = (w.Tg + kSplit - 1) / kSplit;
    int64_t t0 = z * per, t1 = t0 + per < w.Tg ? t0 + per : w.Tg;   // global frames of the slice ...
    t0 -= w.t0g; t1 -= w.t0g;                                       // ... as local rows
    if (s >= w.S || col > w.D) return;
    double acc = 0.0;
    // (unrolled: 16 independent loads in flight per thread; the additions keep their order — t ascending — so the sums keep their bits)
    if (col < w.D) {
#pragma unroll 8
        for (int64_t t = t0; t < t1; ++t) acc += w.gamma[t * w.S + s] * w.rho[t * w.D + col];
    } else {
#pragma unroll 8
        for (int64_t t = t0; t < t1; ++t) acc += w.gamma[t * w.S + s];
    }
    w.rec_out[blockIdx.z * w.stride + static_cast<int64_t>(s) * (w.D + 1) + col] = acc;
}

// last double of a slice record: the sum of the per-frame log-likelihoods of the slice (:623-630), fixed order.  One workgroup per slice.
__global__ __launch_bounds__(kThreads) void vbx_llpart(VbxWs w) {
    __shared__ double red[kThreads];
    const int tid = threadIdx.x;
    const int z = w.z_lo + blockIdx.x;
    const int64_t per = (w.Tg + kSplit - 1) / kSplit;
    int64_t t0 = z * per, t1 = t0 + per < w.Tg ? t0 + per : w.Tg;
    t0 -= w.t0g; t1 -= w.t0g;
    double a = 0.0;
    for (int64_t t = t0 + tid; t < t1; t += kThreads) a += w.llrow[t];
    red[tid] = a;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    if (tid == 0) w.rec_out[blockIdx.x * w.stride + w.stride - 1] = red[0];
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
            for (int z = 0; z < kSplit; ++z) ns += w.rec_in[z * w.stride + static_cast<int64_t>(s) * (D + 1) + D];
            w.pi[s] = ns;
        }
        return;
    }
    double ns = 0.0;
    for (int z = 0; z < kSplit; ++z) ns += w.rec_in[z * w.stride + static_cast<int64_t>(s) * (D + 1) + D];
    const double weight = (w.Fa / w.Fb) * ns;
    double acc = 0.0;
    for (int d = tid; d < D; d += kThreads) {
        double tmp = 0.0;
        for (int z = 0; z < kSplit; ++z) tmp += w.rec_in[z * w.stride + static_cast<int64_t>(s) * (D + 1) + d];
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

__device__ __forceinline__ void vbx_softmax_row(const VbxWs &w, int64_t t, double *g, double mx, int lane);

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
    vbx_softmax_row(w, t, g, mx, lane);
}

// gamma = softmax of the staged row, llrow = its logsumexp (:516-572); `mx` = the row maximum, known to every lane.
__device__ __forceinline__ void vbx_softmax_row(const VbxWs &w, const int64_t t, double *g, const double mx, const int lane) {
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

// ---- many speakers (the hard sessions: AHC leaves hundreds of clusters, e.g. 597 at sigma = 0.041) ------------------------------------------
// Both contractions of an iteration are then real matrix products — gamma^T (rho, 1): [S x T_slice] x [T_slice x (D + 1)] per slice, and
// rho alpha^T: [T x D] x [D x S] — and the kernels above (one speaker per wavefront column / one frame per wavefront, every multiply-add fed by
// two global loads, the alpha rows of 64 lanes in 64 different lines) take 7.6 ms per iteration at S = 597, T = 43 200: 46 ms of the 300 ms
// of that recording.  Tiled form: 64 x 64 outputs per workgroup, 4 x 4 per thread, both operands k-major in LDS (16 k per stage), operands read
// as 16-byte pairs.  Every output is ONE accumulator fed in ascending k by fused multiply-adds with the same operands as the kernels above —
// the same bits, so single-device and sharded runs, and small-S and large-S code paths, agree on every record.  (fp64 vector FMA runs at the
// rate of the fp64 matrix core on this part; the MFMA form would have to keep this summation order to keep the records' bits and does not.)
constexpr int kVT = 64, kVK = 16, kVPad = 2;
constexpr int kVbxTiledMinS = 48;

// record[z][s][0..D] for the slices owned; grid (ceil((D + 1) / 64), ceil(S / 64), z_n)
__global__ __launch_bounds__(kThreads) void vbx_gt_rho_tiled(VbxWs w) {
    __shared__ __attribute__((aligned(16))) double sa[kVK][kVT + kVPad], sb[kVK][kVT + kVPad];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // tx: column quad, ty: speaker quad
    const int c0 = blockIdx.x * kVT, s0 = blockIdx.y * kVT;
    const int z = w.z_lo + blockIdx.z;
    const int64_t per = (w.Tg + kSplit - 1) / kSplit;
    int64_t t0 = z * per, t1 = t0 + per < w.Tg ? t0 + per : w.Tg;
    t0 -= w.t0g; t1 -= w.t0g;
    const int D = w.D, S = w.S;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int64_t k0 = t0; k0 < t1; k0 += kVK) {
        for (int e = tid; e < kVK * kVT; e += kThreads) {
            const int kk = e / kVT, i = e % kVT;
            const int64_t t = k0 + kk;
            const bool in = t < t1;
            sa[kk][i] = in && s0 + i < S ? w.gamma[t * S + s0 + i] : 0.0;
            const int col = c0 + i;
            sb[kk][i] = !in ? 0.0 : (col < D ? w.rho[t * D + col] : (col == D ? 1.0 : 0.0));
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kVK; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = sa[kk][4 * ty + r]; b[r] = sb[kk][4 * tx + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int sp = s0 + 4 * ty + r, col = c0 + 4 * tx + c;
            if (sp < S && col <= D) w.rec_out[blockIdx.z * w.stride + static_cast<int64_t>(sp) * (D + 1) + col] = acc[r][c];
        }
}

// logP staged in gamma: gamma[t][s] = Fa (rho_t . alpha_s - phiT_s / 2 + G_t) + log pi_s; grid (ceil(S / 64), ceil(T / 64))
__global__ __launch_bounds__(kThreads) void vbx_logits_tiled(VbxWs w) {
    __shared__ __attribute__((aligned(16))) double sa[kVK][kVT + kVPad], sb[kVK][kVT + kVPad];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // tx: speaker quad, ty: frame quad
    const int s0 = blockIdx.x * kVT;
    const int64_t f0 = static_cast<int64_t>(blockIdx.y) * kVT;
    const int D = w.D, S = w.S;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int k0 = 0; k0 < D; k0 += kVK) {
        for (int e = tid; e < kVK * kVT; e += kThreads) {
            const int i = e / kVK, kk = e % kVK, d = k0 + kk;       // 16 consecutive d of one row: 128-byte segments
            sa[kk][i] = f0 + i < w.T && d < D ? w.rho[(f0 + i) * D + d] : 0.0;
            sb[kk][i] = s0 + i < S && d < D ? w.alpha[static_cast<int64_t>(s0 + i) * D + d] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kVK; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = sa[kk][4 * ty + r]; b[r] = sb[kk][4 * tx + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t t = f0 + 4 * ty + r;
        if (t >= w.T) continue;
        const double gt = w.G[t];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int sp = s0 + 4 * tx + c;
            if (sp < S) w.gamma[t * S + sp] = fma(fma(w.phiT[sp], -0.5, acc[r][c]) + gt, w.Fa, w.logpi[sp]);
        }
    }
}

// the row soft-max over the staged logits; one wavefront per frame
__global__ __launch_bounds__(kThreads) void vbx_softmax_rows(VbxWs w) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    if (t >= w.T) return;
    double *g = w.gamma + t * w.S;
    double mx = -1.7976931348623157e308;
    for (int s = lane; s < w.S; s += 64) { const double lp = g[s]; mx = lp > mx ? lp : mx; }
    mx = wave_max(mx);
    vbx_softmax_row(w, t, g, mx, lane);
}

// Scalars of one iteration: normalise pi (:605-621), log-likelihood (the slice sums in slice order) and
// ELBO = ll + Fb/2 * sum(log invL - invL - alpha^2 + 1) (:623-647).  One workgroup, fixed order.
__global__ __launch_bounds__(kThreads) void vbx_scalars(VbxWs w) {
    __shared__ double red[kThreads];
    const int tid = threadIdx.x;
    double ll = 0.0;
    for (int z = 0; z < kSplit; ++z) ll += w.rec_in[z * w.stride + w.stride - 1];
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
        const fa_status run = fa::vbx_run_dev(ctx, bX.as<double>(), T, D, blab.as<int32_t>(), S, phi, Fa, Fb, max_iter, epsilon, elbos, n_iters, dev);
        // VBxClustering.refine's catch block (:136-141) covers what runVBx THROWS — an argument its BLAS calls refuse: the refinement degrades to
        // its start, it does not fail.  That is the RUNTIME_ERROR class here (a failing launch, the injected fault).  An allocation failure (a
        // retry may succeed; Swift would not have caught it either) and a refused argument are the caller's to see.
        if (run == FA_ALLOCATION_FAILURE || run == FA_INVALID_ARGUMENT) return run;
        if (run != FA_SUCCESS) {
            const std::string why = ctx->last_error;
            *n_iters = 0;          // elboHistory = []
            FA_TRY(fa::vbx_degrade_dev(ctx, T, S, blab.as<int32_t>(), dev));
            fa::set_error(ctx, FA_SUCCESS, "vbx: degraded to the initial clusters (%s)", why.c_str());
        }
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

// ---- host side -------------------------------------------------------------------------------------------------------------------
namespace {

// buffers + kernel arguments for the frames [t0g, t0g + T) of a problem of Tg frames; the device owns the slices z_lo .. z_lo + z_n - 1
fa_status vbx_setup(fa_ctx *ctx, const double *d_X, int64_t T, int64_t Tg, int64_t t0g, int32_t D, const int32_t *d_labels, int32_t S, const double *phi_host,
                    double Fa, double Fb, int32_t z_lo, int32_t z_n, fa::VbxDevice &o, VbxWs &w) {
    std::vector<double> phic(D);
    for (int d = 0; d < D; ++d) phic[d] = phi_host[d] > 1e-12 ? phi_host[d] : 1e-12;  // :241
    const size_t Tn = static_cast<size_t>(T > 0 ? T : 1);
    const size_t TD = Tn * D, TS = Tn * S, SD = static_cast<size_t>(S) * D;
    const int64_t stride = static_cast<int64_t>(S) * (D + 1) + 1;
    hipError_t e = hipSuccess;
    auto A = [&](fa::DevBuf &b, size_t bytes) { if (e == hipSuccess) e = b.alloc(ctx, bytes); };   // from the context's buffer cache: 13 buffers per refinement
    A(o.phi, 8 * D); A(o.rho, 8 * TD); A(o.G, 8 * Tn); A(o.gamma, 8 * TS); A(o.pi, 8 * S); A(o.logpi, 8 * S);
    A(o.part, 8 * static_cast<size_t>(kSplit) * stride); A(o.alpha, 8 * SD); A(o.invL, 8 * SD); A(o.phiT, 8 * S);
    A(o.ll, 8 * Tn); A(o.scal, 64); A(o.hard, 4 * Tn);
    if (e != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx: device allocation failed"); }
    o.T = T; o.D = D; o.S = S;
    hipStream_t st = ctx->stream;
    FA_HIP_TRY(ctx, hipMemcpyAsync(o.phi.p, phic.data(), 8 * D, hipMemcpyHostToDevice, st));
    FA_HIP_TRY(ctx, hipMemsetAsync(o.ll.p, 0, 8 * Tn, st));   // the records written before the first E-step carry a zero log-likelihood
    w = VbxWs{};
    w.X = d_X; w.phi = o.phi.as<double>(); w.rho = o.rho.as<double>(); w.G = o.G.as<double>();
    w.gamma = o.gamma.as<double>(); w.pi = o.pi.as<double>(); w.logpi = o.logpi.as<double>();
    w.rec_in = o.part.as<double>(); w.rec_out = o.part.as<double>() + static_cast<int64_t>(z_lo) * stride;
    w.alpha = o.alpha.as<double>(); w.invL = o.invL.as<double>(); w.phiT = o.phiT.as<double>(); w.llrow = o.ll.as<double>();
    w.scal = o.scal.as<double>(); w.T = T; w.Tg = Tg; w.t0g = t0g; w.stride = stride; w.D = D; w.S = S; w.z_lo = z_lo; w.z_n = z_n; w.Fa = Fa; w.Fb = Fb;
    w.tiled = !fa::sw_on(fa::Sw::VBX_NO_TILED) ? 1 : 0;   // once per refinement: not inside the iteration (several host threads run refinements at once)
    if (static_cast<size_t>(8) * 4 * D > 64 * 1024) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "vbx: feature dimension too large");
    if (T > 0) {
        const int wave_blocks = static_cast<int>((T + 3) / 4);
        hipLaunchKernelGGL(vbx_prepare, dim3(wave_blocks), dim3(kThreads), 0, st, w);
        hipLaunchKernelGGL(vbx_init_gamma, dim3(wave_blocks), dim3(kThreads), 0, st, w, d_labels, 7.0);
    }
    hipLaunchKernelGGL(vbx_fill, dim3((S + 255) / 256), dim3(256), 0, st, w.pi, S, 1.0 / static_cast<double>(S));  // :239
    FA_HIP_TRY(ctx, hipGetLastError());
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));   // phic (a host temporary) has been consumed
    return FA_SUCCESS;
}

// the records of the owned slices from the present posteriors (and the present per-frame log-likelihoods)
fa_status vbx_records(fa_ctx *ctx, const VbxWs &w) {
    if (w.z_n <= 0) return FA_SUCCESS;
    if (w.S >= kVbxTiledMinS && w.tiled)
        hipLaunchKernelGGL(vbx_gt_rho_tiled, dim3((w.D + 1 + kVT - 1) / kVT, (w.S + kVT - 1) / kVT, w.z_n), dim3(kThreads), 0, ctx->stream, w);
    else
        hipLaunchKernelGGL(vbx_gt_rho, dim3((w.D + 1 + 63) / 64, (w.S + 3) / 4, w.z_n), dim3(kThreads), 0, ctx->stream, w);
    hipLaunchKernelGGL(vbx_llpart, dim3(w.z_n), dim3(kThreads), 0, ctx->stream, w);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}
// speaker statistics from the complete records, then the E-step of the frames held here (:330-572)
fa_status vbx_estep_phase(fa_ctx *ctx, const VbxWs &w) {
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(vbx_speaker, dim3(w.S), dim3(kThreads), 0, st, w, 0);
    hipLaunchKernelGGL(vbx_logpi, dim3((w.S + 255) / 256), dim3(256), 0, st, w);
    if (w.T > 0 && w.S >= kVbxTiledMinS && w.tiled) {
        hipLaunchKernelGGL(vbx_logits_tiled, dim3((w.S + kVT - 1) / kVT, static_cast<unsigned>((w.T + kVT - 1) / kVT)), dim3(kThreads), 0, st, w);
        hipLaunchKernelGGL(vbx_softmax_rows, dim3(static_cast<int>((w.T + 3) / 4)), dim3(kThreads), 0, st, w);
    } else if (w.T > 0)
        hipLaunchKernelGGL(vbx_estep, dim3(static_cast<int>((w.T + 3) / 4)), dim3(kThreads), sizeof(double) * 4 * static_cast<size_t>(w.D), st, w);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}
// pi of the new posteriors (column D of the complete records, :586-621) and the ELBO (:623-647)
fa_status vbx_finish_phase(fa_ctx *ctx, const VbxWs &w, double *elbo) {
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(vbx_speaker, dim3(w.S), dim3(kThreads), 0, st, w, 1);
    hipLaunchKernelGGL(vbx_scalars, dim3(1), dim3(kThreads), 0, st, w);
    FA_HIP_TRY(ctx, hipGetLastError());
    FA_HIP_TRY(ctx, hipMemcpyAsync(elbo, w.scal, sizeof(double), hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(ctx, hipStreamSynchronize(st));
    return FA_SUCCESS;
}
fa_status vbx_hard_phase(fa_ctx *ctx, const VbxWs &w, int32_t *d_hard) {
    if (w.T > 0) hipLaunchKernelGGL(vbx_hard, dim3(static_cast<int>((w.T + 255) / 256)), dim3(256), 0, ctx->stream, w, d_hard);
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

}  // namespace

// The EM loop on device-resident inputs (d_X: [T][D] rho features, d_labels: [T] AHC labels with S distinct values); gamma, pi and
// the hard assignment stay on the device in `out`.  The ELBO of every iteration crosses to the host (8 bytes) for the
// convergence test of the reference (:653-659).
fa_status fa::vbx_run_dev(fa_ctx *ctx, const double *d_X, int64_t T, int32_t D, const int32_t *d_labels, int32_t S, const double *phi_host,
                          double Fa, double Fb, int32_t max_iter, double epsilon, double *elbos, int32_t *n_iters, fa::VbxDevice &o) {
    if (n_iters) *n_iters = 0;
    VbxWs w;
    FA_TRY(vbx_setup(ctx, d_X, T, T, 0, D, d_labels, S, phi_host, Fa, Fb, 0, kSplit, o, w));
    if (fa::fault_hit(FA_FAULT_VBX)) return fa::set_error(ctx, FA_RUNTIME_ERROR, "vbx: injected failure");
    FA_TRY(vbx_records(ctx, w));
    double prev = -1.7976931348623157e308;
    int iters = 0;
    for (int it = 0; it < max_iter; ++it) {
        iters = it + 1;
        FA_TRY(vbx_estep_phase(ctx, w));
        FA_TRY(vbx_records(ctx, w));      // of the NEW posteriors: pi and the log-likelihood of this iteration, the statistics of the next
        double elbo = 0.0;
        FA_TRY(vbx_finish_phase(ctx, w, &elbo));
        if (elbos) elbos[it] = elbo;
        if (it > 0 && std::fabs(elbo - prev) < epsilon) { prev = elbo; break; }  // :653-659
        prev = elbo;
    }
    FA_TRY(vbx_hard_phase(ctx, w, o.hard.as<int32_t>()));
    if (n_iters) *n_iters = iters;
    return FA_SUCCESS;
}

// VBxClustering.refine's catch block (VBxClustering.swift:136-141): when runVBx throws, the refinement does not fail — it returns
// gamma = initialGamma (the plain one-hot of the clamped initial labels, :100-104, NOT the smoothed start of runVBx), pi = 1/S, no ELBOs,
// and hardClusters = argmax of that gamma = the clamped labels (:144-146).  The stages behind it go on with those.  Buffers of `o` that a
// failed run left allocated are reused; the three outputs are (re)allocated if the failure was the allocation itself.
__global__ void vbx_degrade_kernel(const int32_t *__restrict__ labels, double *__restrict__ gamma, int32_t *__restrict__ hard, const int64_t T, const int32_t S) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= T * S) return;
    const int64_t t = i / S;
    const int32_t s = static_cast<int32_t>(i - t * S);
    int32_t l = labels[t];
    l = l < 0 ? 0 : (l > S - 1 ? S - 1 : l);   // max(0, min(cluster, speakerCount - 1)) (:102)
    gamma[i] = s == l ? 1.0 : 0.0;
    if (s == 0) hard[t] = l;
}
fa_status fa::vbx_degrade_dev(fa_ctx *ctx, int64_t T, int32_t S, const int32_t *d_labels, fa::VbxDevice &o) {
    (void)hipGetLastError();
    if (T < 0 || S < 1) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "vbx degrade: bad shape");
    const size_t Tn = static_cast<size_t>(T > 0 ? T : 1);
    hipError_t e = hipSuccess;
    auto need = [&](fa::DevBuf &b, size_t bytes) { if (e == hipSuccess && (!b.p || b.cap < bytes)) { b.reset(); e = b.alloc(ctx, bytes); } };
    need(o.gamma, 8 * Tn * S); need(o.pi, 8 * static_cast<size_t>(S)); need(o.hard, 4 * Tn);
    if (e != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx degrade: device allocation failed"); }
    o.T = T; o.S = S;
    if (T > 0) {
        const int64_t total = T * S;
        hipLaunchKernelGGL(vbx_degrade_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, ctx->stream, d_labels, o.gamma.as<double>(),
                           o.hard.as<int32_t>(), T, S);
    }
    hipLaunchKernelGGL(vbx_fill, dim3((S + 255) / 256), dim3(256), 0, ctx->stream, o.pi.as<double>(), S, 1.0 / static_cast<double>(S));   // :139
    FA_HIP_TRY(ctx, hipGetLastError());
    return FA_SUCCESS;
}

// ---- sharded over the frame axis -------------------------------------------------------------------------------
// One fa_vbx_shard per device holds a contiguous range of the 64 slices (world sizes that divide 64).  The caller moves the records:
//   begin(chunk) -> all-gather chunks -> repeat { iterate(full, chunk) -> all-gather -> finish_iteration(full, &elbo) } -> result.
// Every device evaluates the speaker statistics, pi and the ELBO from the same complete records, so all of them take the same
// convergence decision without a broadcast; the only collective is the all-gather of 64 (S (D + 1) + 1) doubles per iteration.
struct fa_vbx_shard {
    fa_ctx *ctx = nullptr;
    fa::VbxDevice dev;
    fa::DevBuf X, labels;
    VbxWs w{};
    int32_t rank = 0, world = 1;
    int64_t t_lo = 0, t_hi = 0;
};

extern "C" {

int32_t fa_vbx_shard_slices(void) { return kSplit; }

void fa_vbx_shard_range(int64_t T_total, int32_t rank, int32_t world, int64_t *t_lo, int64_t *t_hi) {
    int64_t lo = 0, hi = 0;
    if (T_total > 0 && world > 0 && kSplit % world == 0 && rank >= 0 && rank < world) {
        const int64_t per = (T_total + kSplit - 1) / kSplit, zn = kSplit / world;
        lo = rank * zn * per; hi = (rank + 1) * zn * per;
        lo = lo < T_total ? lo : T_total; hi = hi < T_total ? hi : T_total;
    }
    if (t_lo) *t_lo = lo;
    if (t_hi) *t_hi = hi;
}

int64_t fa_vbx_shard_chunk_doubles(int32_t S, int32_t D, int32_t world) {
    if (S < 1 || D < 1 || world < 1 || kSplit % world != 0) return 0;
    return (kSplit / world) * (static_cast<int64_t>(S) * (D + 1) + 1);
}

fa_status fa_vbx_shard_create(fa_ctx *ctx, const double *rho_local, int64_t T_total, int32_t D, const int32_t *labels_local, int32_t S,
                              const double *phi, double Fa, double Fb, int32_t rank, int32_t world, fa_vbx_shard **out) {
    if (!ctx || !out) return FA_INVALID_ARGUMENT;
    *out = nullptr;
    if (T_total <= 0 || D <= 0 || S < 1 || !phi || world < 1 || kSplit % world != 0 || rank < 0 || rank >= world)
        return fa::set_error(ctx, FA_INVALID_ARGUMENT, "vbx shard: bad shape (the world size must divide 64)");
    fa::DeviceGuard guard(ctx->device);
    try {
        fa_vbx_shard *h = new fa_vbx_shard;
        h->ctx = ctx; h->rank = rank; h->world = world;
        fa_vbx_shard_range(T_total, rank, world, &h->t_lo, &h->t_hi);
        const int64_t T = h->t_hi - h->t_lo;
        if (T > 0 && (!rho_local || !labels_local)) { delete h; return FA_INVALID_ARGUMENT; }
        const size_t Tn = static_cast<size_t>(T > 0 ? T : 1);
        if (h->X.alloc(8 * Tn * D) != hipSuccess || h->labels.alloc(4 * Tn) != hipSuccess) {
            (void)hipGetLastError(); delete h;
            return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx shard: device allocation failed");
        }
        fa_status st = FA_SUCCESS;
        if (T > 0) {
            hipError_t e = hipMemcpyAsync(h->X.p, rho_local, 8 * static_cast<size_t>(T) * D, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h->labels.p, labels_local, 4 * static_cast<size_t>(T), hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) st = fa::hip_status(ctx, e, "vbx shard upload");
        }
        const int32_t zn = kSplit / world;
        if (st == FA_SUCCESS) st = vbx_setup(ctx, h->X.as<double>(), T, T_total, h->t_lo, D, h->labels.as<int32_t>(), S, phi, Fa, Fb, rank * zn, zn, h->dev, h->w);
        if (st != FA_SUCCESS) { delete h; return st; }
        *out = h;
        return FA_SUCCESS;
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "vbx shard: host allocation failed");
    } catch (...) {
        return fa::set_error(ctx, FA_UNKNOWN_ERROR, "vbx shard: unexpected failure");
    }
}

void fa_vbx_shard_destroy(fa_vbx_shard *h) {
    if (!h) return;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
}

void fa_vbx_shard_frames(const fa_vbx_shard *h, int64_t *t_lo, int64_t *t_hi) {
    if (t_lo) *t_lo = h ? h->t_lo : 0;
    if (t_hi) *t_hi = h ? h->t_hi : 0;
}

// d_chunk: DEVICE double[fa_vbx_shard_chunk_doubles]: the records of the slices held here, from the initial posteriors.  Complete on return.
fa_status fa_vbx_shard_begin(fa_vbx_shard *h, double *d_chunk) {
    if (!h || !d_chunk) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(h->ctx->device);
    VbxWs w = h->w;
    w.rec_out = d_chunk;
    FA_TRY(vbx_records(h->ctx, w));
    FA_HIP_TRY(h->ctx, hipStreamSynchronize(h->ctx->stream));
    return FA_SUCCESS;
}

// d_full: DEVICE double[64 x record]: the gathered records of the present posteriors; d_chunk: the records of the new ones.  Complete on return.
fa_status fa_vbx_shard_iterate(fa_vbx_shard *h, const double *d_full, double *d_chunk) {
    if (!h || !d_full || !d_chunk) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(h->ctx->device);
    VbxWs w = h->w;
    w.rec_in = d_full; w.rec_out = d_chunk;
    FA_TRY(vbx_estep_phase(h->ctx, w));
    FA_TRY(vbx_records(h->ctx, w));
    FA_HIP_TRY(h->ctx, hipStreamSynchronize(h->ctx->stream));
    return FA_SUCCESS;
}

// d_full: the gathered records of the posteriors fa_vbx_shard_iterate just wrote; *elbo: the ELBO of the iteration (:623-647)
fa_status fa_vbx_shard_finish_iteration(fa_vbx_shard *h, const double *d_full, double *elbo) {
    if (!h || !d_full || !elbo) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(h->ctx->device);
    VbxWs w = h->w;
    w.rec_in = d_full;
    return vbx_finish_phase(h->ctx, w, elbo);
}

// HOST outputs: gamma_local [frames held][S], pi [S], hard_local [frames held] (each may be NULL)
fa_status fa_vbx_shard_result(fa_vbx_shard *h, double *gamma_local, double *pi, int32_t *hard_local) {
    if (!h) return FA_INVALID_ARGUMENT;
    fa_ctx *ctx = h->ctx;
    fa::DeviceGuard guard(ctx->device);
    const int64_t T = h->t_hi - h->t_lo;
    FA_TRY(vbx_hard_phase(ctx, h->w, h->dev.hard.as<int32_t>()));
    if (gamma_local && T > 0) FA_HIP_TRY(ctx, hipMemcpyAsync(gamma_local, h->dev.gamma.p, 8 * static_cast<size_t>(T) * h->w.S, hipMemcpyDeviceToHost, ctx->stream));
    if (pi) FA_HIP_TRY(ctx, hipMemcpyAsync(pi, h->dev.pi.p, 8 * static_cast<size_t>(h->w.S), hipMemcpyDeviceToHost, ctx->stream));
    if (hard_local && T > 0) FA_HIP_TRY(ctx, hipMemcpyAsync(hard_local, h->dev.hard.p, 4 * static_cast<size_t>(T), hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FA_SUCCESS;
}

}  // extern "C"
