// offline.hip — the clustering stage of the offline diarizer as ONE device-resident call.
//
// Replaces the arithmetic of OfflineDiarizerManager.cluster
// (reference: Sources/FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:270-375) on precomputed embeddings:
//   selectTrainingEmbeddings (:591-611) -> AHCClustering.cluster (threshold, :301-306) -> VBxClustering.refineWithConstraints
//   (:308-333) -> computeCentroids (:613-691, fallback computeCentroidsFromClusters :693-740) -> centroid scores + constrained
//   per-chunk assignment, or the plain cosine argmax (:345-375, :789-822).
// The embeddings (fp32, widened to fp64 on the device like `embeddingFeatures.map { $0.map(Double.init) }`, :286) and the PLDA
// features go up ONCE; between the stages only what the host has to decide on crosses PCIe: the dendrogram (32 bytes per merge)
// for the O(N) cut, the label vector back, 8 bytes of ELBO per VBx iteration, pi (S doubles), and the final labels.  The
// stages themselves are the device cores the single-stage entries use (fa_common.h), so every intermediate result equals the
// one the stage-by-stage Python glue of round 1 produced.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "fa_common.h"

namespace {

__global__ void widen_rows(const float *__restrict__ x, const int32_t *__restrict__ rows, double *__restrict__ out, int64_t n_out, int d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_out * d) return;
    const int64_t r = i / d, k = i - r * d;
    const int64_t src = rows ? rows[r] : r;
    out[i] = static_cast<double>(x[src * d + k]);
}

__global__ void gather_rows_f64(const double *__restrict__ x, const int32_t *__restrict__ rows, double *__restrict__ out, int64_t n_out, int d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_out * d) return;
    const int64_t r = i / d, k = i - r * d;
    out[i] = x[static_cast<int64_t>(rows[r]) * d + k];
}

// one flag per row: every element finite (selectTrainingEmbeddings keeps rows without NaN / Inf)
__global__ void finite_rows(const float *__restrict__ x, uint8_t *__restrict__ ok, int64_t n, int d) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    int good = 1;
    for (int k = lane; k < d; k += 64) good &= isfinite(x[r * d + k]) ? 1 : 0;
    const unsigned long long all = __ballot(good != 0);
    if (lane == 0) ok[r] = all == ~0ull ? 1 : 0;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

namespace {

// One recording's pass through the stage, split where the merge chains of several recordings can run together
// (fa_offline_cluster_batch): prepare() = inputs + training rows + normalised rows on the device; the caller runs the linkage
// (alone or batched); finish() = cut, VBx, centroids, assignment.  Everything is enqueued on the context's stream.
struct ClusterJob {
    fa_ctx *ctx;
    const float *embeddings; int64_t n; int32_t d; const double *rho; int32_t rho_dim; const int32_t *chunk_indices; const double *phi;
    const fa_offline_cluster_config *config; int32_t device_pointers;
    int32_t *labels; double *centroids; int32_t max_centroids; int32_t *n_centroids; fa_offline_cluster_info *info;
    // optional copies of the intermediates (fa_offline_cluster_ex): AHC labels and VBx hard labels of the training rows, ELBO per iteration
    int32_t *aux_ahc = nullptr; int32_t *aux_hard = nullptr; double *aux_elbos = nullptr;

    fa::DevBuf b_emb32, b_rho_in, b_ok, b_emb, b_temb, b_trho, b_train, b_norm, b_z;
    const float *d_emb32 = nullptr;
    const double *d_rho_all = nullptr, *d_temb = nullptr, *d_trho = nullptr;
    int64_t nt = 0;
    bool rows_finite = false;   // every training row is free of NaN / Inf (the centroid sums may then add a zero-weight row instead of skipping it: same bits)
    double t_begin = 0, t_inputs = 0, t_ahc = 0;
    fa_ahc_stats ahc_stats{};

    fa_status check_args() {
        if (!ctx || !config || !labels || !n_centroids) return FA_INVALID_ARGUMENT;
        *n_centroids = 0;
        if (info) memset(info, 0, sizeof(*info));
        if (n <= 0) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "offline cluster: no embeddings (noSpeechDetected, :281-283)");
        if (n > INT32_MAX || d < 1 || rho_dim < 0 || !embeddings || (rho_dim > 0 && (!rho || !phi)) || (config->constrained_assignment && !chunk_indices))
            return fa::set_error(ctx, FA_INVALID_ARGUMENT, "offline cluster: bad arguments");
        return FA_SUCCESS;
    }

    // inputs to the device (once), selectTrainingEmbeddings, unit rows for the linkage (b_norm) and room for the dendrogram (b_z) when nt >= 2
    fa_status prepare() {
        hipStream_t st = ctx->stream;
        t_begin = now_s();
        d_emb32 = embeddings;
        d_rho_all = rho;
        if (!device_pointers) {
            if (b_emb32.alloc(ctx, sizeof(float) * n * d) != hipSuccess || (rho_dim > 0 && b_rho_in.alloc(ctx, sizeof(double) * n * rho_dim) != hipSuccess)) {
                (void)hipGetLastError();
                return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: input allocation failed");
            }
            FA_HIP_TRY(ctx, hipMemcpyAsync(b_emb32.p, embeddings, sizeof(float) * n * d, hipMemcpyHostToDevice, st));
            if (rho_dim > 0) FA_HIP_TRY(ctx, hipMemcpyAsync(b_rho_in.p, rho, sizeof(double) * n * rho_dim, hipMemcpyHostToDevice, st));
            d_emb32 = b_emb32.as<float>();
            d_rho_all = b_rho_in.as<double>();
        }
        // ---- selectTrainingEmbeddings (:591-611): rows without NaN / Inf; all rows if none qualifies
        if (b_ok.alloc(ctx, n) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
        hipLaunchKernelGGL(finite_rows, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, st, d_emb32, b_ok.as<uint8_t>(), n, d);
        FA_HIP_TRY(ctx, hipGetLastError());
        std::vector<uint8_t> ok(static_cast<size_t>(n));
        FA_HIP_TRY(ctx, hipMemcpyAsync(ok.data(), b_ok.p, n, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));
        std::vector<int32_t> train;
        for (int64_t i = 0; i < n; ++i) if (ok[i]) train.push_back(static_cast<int32_t>(i));
        rows_finite = !train.empty();   // the training rows are the finite ones — unless none is, and all rows train (:606-609)
        const bool all_rows = train.empty() || static_cast<int64_t>(train.size()) == n;
        if (train.empty()) { train.resize(n); for (int64_t i = 0; i < n; ++i) train[i] = static_cast<int32_t>(i); }
        nt = static_cast<int64_t>(train.size());
        if (b_emb.alloc(ctx, sizeof(double) * n * d) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
        const unsigned g_all = static_cast<unsigned>((n * d + 255) / 256);
        hipLaunchKernelGGL(widen_rows, dim3(g_all), dim3(256), 0, st, d_emb32, static_cast<const int32_t *>(nullptr), b_emb.as<double>(), n, d);   // Float -> Double (:286)
        d_temb = b_emb.as<double>();
        d_trho = d_rho_all;
        if (!all_rows) {
            if (b_train.alloc(ctx, sizeof(int32_t) * nt) != hipSuccess || b_temb.alloc(ctx, sizeof(double) * nt * d) != hipSuccess ||
                (rho_dim > 0 && b_trho.alloc(ctx, sizeof(double) * nt * rho_dim) != hipSuccess)) {
                (void)hipGetLastError();
                return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed");
            }
            FA_HIP_TRY(ctx, hipMemcpyAsync(b_train.p, train.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(gather_rows_f64, dim3(static_cast<unsigned>((nt * d + 255) / 256)), dim3(256), 0, st, b_emb.as<double>(), b_train.as<int32_t>(), b_temb.as<double>(), nt, d);
            if (rho_dim > 0)
                hipLaunchKernelGGL(gather_rows_f64, dim3(static_cast<unsigned>((nt * rho_dim + 255) / 256)), dim3(256), 0, st, d_rho_all, b_train.as<int32_t>(), b_trho.as<double>(), nt, rho_dim);
            d_temb = b_temb.as<double>();
            d_trho = b_trho.as<double>();
        }
        FA_HIP_TRY(ctx, hipGetLastError());
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));   // `train` is a host temporary
        if (nt >= 2) {   // AHC input (:301-306): unit rows
            if (b_norm.alloc(ctx, sizeof(double) * nt * d) != hipSuccess || b_z.alloc(ctx, sizeof(double) * 4 * (nt - 1)) != hipSuccess) {
                (void)hipGetLastError();
                return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed");
            }
            FA_TRY(fa::ahc_normalize_dev(ctx, d_temb, b_norm.as<double>(), nt, d));
        }
        t_inputs = now_s();
        return FA_SUCCESS;
    }

    // ahc_status: what the linkage of b_norm into b_z returned (ignored when nt < 2)
    fa_status finish(const fa_status ahc_status) {
        hipStream_t st = ctx->stream;
        // ---- cut (:301-306); fewer than 2 training rows -> all 0; a failed linkage degrades to singletons (AHCClustering.swift:52-55)
        std::vector<int32_t> initial(static_cast<size_t>(nt), 0);
        if (nt >= 2) {
            if (ahc_status != FA_SUCCESS) {
                for (int64_t i = 0; i < nt; ++i) initial[i] = static_cast<int32_t>(i);
            } else {
                std::vector<double> z(static_cast<size_t>(4 * (nt - 1)));
                FA_HIP_TRY(ctx, hipMemcpyAsync(z.data(), b_z.p, sizeof(double) * z.size(), hipMemcpyDeviceToHost, st));
                FA_HIP_TRY(ctx, hipStreamSynchronize(st));
                FA_TRY(fa_ahc_cut(z.data(), static_cast<size_t>(nt), config->clustering_threshold, initial.data()));
            }
        }
        if (aux_ahc) memcpy(aux_ahc, initial.data(), sizeof(int32_t) * static_cast<size_t>(nt));
        t_ahc = now_s();
        // ---- VBx (:308-333)
        const int32_t S = nt > 0 ? std::max(1, fa_vbx_speaker_count(initial.data(), nt)) : 0;
        fa::VbxDevice vbx;
        fa::DevBuf b_lab;
        std::vector<double> pi;
        std::vector<int32_t> hard;
        bool have_vbx = false, adjusted = false, vbx_degraded = false;
        int32_t vbx_iters = 0;
        std::vector<double> km_centroids;
        std::vector<int32_t> km_labels;
        int32_t km_k = 0;
        if (rho_dim > 0 && nt > 0) {
            if (b_lab.alloc(ctx, sizeof(int32_t) * nt) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
            FA_HIP_TRY(ctx, hipMemcpyAsync(b_lab.p, initial.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice, st));
            std::vector<double> elbos(static_cast<size_t>(std::max(config->max_vbx_iterations, 1)));
            const fa_status vbx_st = fa::vbx_run_dev(ctx, d_trho, nt, rho_dim, b_lab.as<int32_t>(), S, phi, config->warm_start_fa, config->warm_start_fb,
                                                     config->max_vbx_iterations, config->convergence_tolerance, elbos.data(), &vbx_iters, vbx);
            if (vbx_st == FA_ALLOCATION_FAILURE || vbx_st == FA_INVALID_ARGUMENT) return vbx_st;   // not what the reference's catch covers (see fa_vbx_refine)
            if (vbx_st != FA_SUCCESS) {   // VBxClustering.refine's catch block (VBxClustering.swift:136-141): gamma = one-hot AHC labels, pi = 1/S, no ELBOs — and on
                const std::string why = ctx->last_error;
                vbx_iters = 0;
                vbx_degraded = true;
                FA_TRY(fa::vbx_degrade_dev(ctx, nt, S, b_lab.as<int32_t>(), vbx));
                fa::set_error(ctx, FA_SUCCESS, "offline cluster: VBx degraded to the AHC clusters (%s)", why.c_str());   // a SUCCESS return does not leave a failure text behind
            }
            pi.resize(S);
            FA_HIP_TRY(ctx, hipMemcpyAsync(pi.data(), vbx.pi.p, sizeof(double) * S, hipMemcpyDeviceToHost, st));
            const bool has_constraints = config->num_speakers >= 0 || config->min_speakers >= 0 || config->max_speakers >= 0;   // :309-312
            if (has_constraints) {
                hard.resize(nt);
                FA_HIP_TRY(ctx, hipMemcpyAsync(hard.data(), vbx.hard.p, sizeof(int32_t) * nt, hipMemcpyDeviceToHost, st));
            }
            if (aux_hard) FA_HIP_TRY(ctx, hipMemcpyAsync(aux_hard, vbx.hard.p, sizeof(int32_t) * nt, hipMemcpyDeviceToHost, st));
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (aux_elbos) memcpy(aux_elbos, elbos.data(), sizeof(double) * static_cast<size_t>(std::max(vbx_iters, 0)));
            have_vbx = true;
            if (has_constraints) {   // refineWithConstraints (VBxClustering.swift:685-733)
                const int64_t ns = config->num_speakers, mn = config->min_speakers, mx = config->max_speakers;
                int64_t res[3];
                fa_speaker_constraints_resolve(nt, ns >= 0 ? &ns : nullptr, mn >= 0 ? &mn : nullptr, mx >= 0 ? &mx : nullptr, res);
                std::vector<int32_t> used(hard);
                std::sort(used.begin(), used.end());
                const int64_t detected = std::unique(used.begin(), used.end()) - used.begin();   // assignedClusterCount (OfflineDiarizerTypes.swift:687-702)
                if (detected < res[1] || detected > res[2]) {
                    const int32_t target = static_cast<int32_t>(std::min(std::max(detected, res[1]), res[2]));
                    std::vector<double> temb_host(static_cast<size_t>(nt) * d);   // the fallback is rare: it takes the host-pointer K-Means entry
                    FA_HIP_TRY(ctx, hipMemcpyAsync(temb_host.data(), d_temb, sizeof(double) * nt * d, hipMemcpyDeviceToHost, st));
                    FA_HIP_TRY(ctx, hipStreamSynchronize(st));
                    km_labels.resize(nt);
                    km_centroids.assign(static_cast<size_t>(std::max<int64_t>(std::min<int64_t>(target, nt), 1)) * d, 0.0);
                    FA_TRY(fa_kmeans_cluster_ninit(ctx, temb_host.data(), nt, d, target, 100, 10, 0, km_labels.data(), km_centroids.data(), &km_k, nullptr, nullptr));
                    adjusted = true;
                }
            }
        }
        const double t_vbx = now_s();

        // ---- centroids (:613-691): K-Means centroids as they are (:622-629), else gamma-weighted means of the speakers with pi > 1e-7,
        //      else per-cluster means of the AHC labels (computeCentroidsFromClusters, sequential sums)
        fa::DevBuf b_cent, b_spk;
        int32_t K = 0;
        if (adjusted && km_k > 0) {
            K = km_k;
            if (b_cent.alloc(ctx, sizeof(double) * K * d) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
            FA_HIP_TRY(ctx, hipMemcpyAsync(b_cent.p, km_centroids.data(), sizeof(double) * K * d, hipMemcpyHostToDevice, st));
        } else if (have_vbx) {
            std::vector<int32_t> spk;
            for (int s = 0; s < S; ++s) if (pi[s] > 1e-7) spk.push_back(s);
            K = static_cast<int32_t>(spk.size());
            if (K > 0) {
                if (b_cent.alloc(ctx, sizeof(double) * K * d) != hipSuccess || b_spk.alloc(ctx, sizeof(int32_t) * K) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
                FA_HIP_TRY(ctx, hipMemcpyAsync(b_spk.p, spk.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, st));
                FA_TRY(fa::centroids_dev(ctx, d_temb, nt, d, vbx.gamma.as<double>(), S, b_spk.as<int32_t>(), K, b_cent.as<double>(), rows_finite));
                FA_HIP_TRY(ctx, hipStreamSynchronize(st));   // spk is a host temporary
            }
        }
        if (K == 0 && nt > 0) {
            std::vector<double> temb_host(static_cast<size_t>(nt) * d);
            FA_HIP_TRY(ctx, hipMemcpyAsync(temb_host.data(), d_temb, sizeof(double) * nt * d, hipMemcpyDeviceToHost, st));
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
            int32_t kmax = 0;
            for (int64_t i = 0; i < nt; ++i) kmax = std::max(kmax, initial[i] + 1);
            std::vector<double> sum(static_cast<size_t>(kmax) * d, 0.0);
            std::vector<int64_t> cnt(kmax, 0);
            for (int64_t i = 0; i < nt; ++i) {
                ++cnt[initial[i]];
                for (int k = 0; k < d; ++k) sum[static_cast<size_t>(initial[i]) * d + k] += temb_host[i * d + k];
            }
            std::vector<double> cen;
            for (int c = 0; c < kmax; ++c) if (cnt[c] > 0) for (int k = 0; k < d; ++k) cen.push_back(sum[static_cast<size_t>(c) * d + k] / static_cast<double>(cnt[c]));
            K = static_cast<int32_t>(cen.size() / d);
            if (b_cent.alloc(ctx, sizeof(double) * std::max(K, 1) * d) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
            FA_HIP_TRY(ctx, hipMemcpyAsync(b_cent.p, cen.data(), sizeof(double) * cen.size(), hipMemcpyHostToDevice, st));
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
        }

        // ---- assignment of ALL embeddings (:345-375): constrained per chunk unless the count was forced or there is a single centroid
        fa::DevBuf b_cn, b_scores, b_out;
        if (b_cn.alloc(ctx, sizeof(double) * std::max(K, 1) * d) != hipSuccess || b_out.alloc(ctx, sizeof(int32_t) * n) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
        const bool constrained = config->constrained_assignment && !adjusted && K > 1;   // :355-358
        if (constrained) {
            if (b_scores.alloc(ctx, sizeof(double) * n * K) != hipSuccess) { (void)hipGetLastError(); return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: allocation failed"); }
            FA_TRY(fa::scores_dev(ctx, b_emb.as<double>(), n, d, b_cent.as<double>(), K, b_cn.as<double>(), b_scores.as<double>()));
            FA_TRY(fa::constrained_assign_dev(ctx, b_scores.as<double>(), n, K, chunk_indices, b_out.as<int32_t>()));
        } else {
            FA_TRY(fa::assign_dev(ctx, b_emb.as<double>(), n, d, b_cent.as<double>(), K, b_cn.as<double>(), b_out.as<int32_t>()));
        }
        *n_centroids = K;
        if (centroids && K > max_centroids) {   // checked BEFORE any output copy is enqueued: on this error the caller's buffers are untouched
            FA_HIP_TRY(ctx, hipStreamSynchronize(st));
            return fa::set_error(ctx, FA_OUTPUT_TOO_SMALL, "offline cluster: %d centroids, room for %d", K, max_centroids);
        }
        FA_HIP_TRY(ctx, hipMemcpyAsync(labels, b_out.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        if (centroids && K > 0) FA_HIP_TRY(ctx, hipMemcpyAsync(centroids, b_cent.p, sizeof(double) * K * d, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(ctx, hipStreamSynchronize(st));
        const double t_end = now_s();
        if (info) {
            info->training_rows = nt; info->initial_clusters = S; info->vbx_iterations = vbx_iters; info->was_adjusted = adjusted ? 1 : 0;
            info->constrained = constrained ? 1 : 0; info->vbx_degraded = vbx_degraded ? 1 : 0; info->ahc_degraded = (nt >= 2 && ahc_status != FA_SUCCESS) ? 1 : 0;
            info->inputs_s = t_inputs - t_begin; info->ahc_s = t_ahc - t_inputs; info->vbx_s = t_vbx - t_ahc; info->assign_s = t_end - t_vbx;
            info->total_s = t_end - t_begin; info->ahc = ahc_stats;
        }
        return FA_SUCCESS;
    }
};

template <class F>
fa_status guarded(fa_ctx *ctx, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fa::set_error(ctx, FA_ALLOCATION_FAILURE, "offline cluster: host allocation failed");
    } catch (...) {
        return fa::set_error(ctx, FA_UNKNOWN_ERROR, "offline cluster: unexpected failure");
    }
}

}  // namespace

extern "C" {

void fa_offline_cluster_default_config(fa_offline_cluster_config *c) {
    if (!c) return;
    c->clustering_threshold = 0.6; c->warm_start_fa = 0.07; c->warm_start_fb = 0.8;     // OfflineDiarizerTypes.swift:155-163,189-192
    c->max_vbx_iterations = 20; c->convergence_tolerance = 1e-4; c->constrained_assignment = 1;
    c->num_speakers = -1; c->min_speakers = -1; c->max_speakers = -1; c->ahc_mode = FA_AHC_MODE_AUTO;
}

fa_status fa_offline_cluster(fa_ctx *ctx, const float *embeddings, int64_t n, int32_t d, const double *rho, int32_t rho_dim,
                             const int32_t *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                             int32_t device_pointers, int32_t *labels, double *centroids, int32_t max_centroids, int32_t *n_centroids,
                             fa_offline_cluster_info *info) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    ClusterJob job{ctx, embeddings, n, d, rho, rho_dim, chunk_indices, phi, config, device_pointers, labels, centroids, max_centroids, n_centroids, info};
    FA_TRY(job.check_args());
    fa::DeviceGuard guard(ctx->device);
    return guarded(ctx, [&]() -> fa_status {
        FA_TRY(job.prepare());
        fa_status ahc_st = FA_SUCCESS;
        if (job.nt >= 2)
            ahc_st = fa::fault_hit(FA_FAULT_AHC) ? FA_RUNTIME_ERROR
                                                 : fa::ahc_run_device(ctx, job.b_norm.as<double>(), static_cast<size_t>(job.nt), static_cast<size_t>(d), job.b_z.as<double>(), config->ahc_mode, &job.ahc_stats);
        return job.finish(ahc_st);
    });
}

// fa_offline_cluster + copies of the stage's intermediates for verification at full size (bench.py and the 8 h digest test compare
// them with the CPU side): ahc_labels [training rows] = AHCClustering.cluster's output, vbx_hard [training rows] = argmax of gamma,
// elbos [max_vbx_iterations] (info->vbx_iterations of them are written).  Every pointer may be NULL.
fa_status fa_offline_cluster_ex(fa_ctx *ctx, const float *embeddings, int64_t n, int32_t d, const double *rho, int32_t rho_dim,
                                const int32_t *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                                int32_t device_pointers, int32_t *labels, double *centroids, int32_t max_centroids, int32_t *n_centroids,
                                fa_offline_cluster_info *info, int32_t *ahc_labels, int32_t *vbx_hard, double *elbos) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    ClusterJob job{ctx, embeddings, n, d, rho, rho_dim, chunk_indices, phi, config, device_pointers, labels, centroids, max_centroids, n_centroids, info,
                   ahc_labels, vbx_hard, elbos};
    FA_TRY(job.check_args());
    fa::DeviceGuard guard(ctx->device);
    return guarded(ctx, [&]() -> fa_status {
        FA_TRY(job.prepare());
        fa_status ahc_st = FA_SUCCESS;
        if (job.nt >= 2)
            ahc_st = fa::fault_hit(FA_FAULT_AHC) ? FA_RUNTIME_ERROR
                                                 : fa::ahc_run_device(ctx, job.b_norm.as<double>(), static_cast<size_t>(job.nt), static_cast<size_t>(d), job.b_z.as<double>(), config->ahc_mode, &job.ahc_stats);
        return job.finish(ahc_st);
    });
}

}  // extern "C"

namespace {
fa_status cluster_batch(fa_ctx *ctx, int32_t count, const float *const *embeddings, const int64_t *n, int32_t d, const double *const *rho,
                        int32_t rho_dim, const int32_t *const *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                        const int32_t device_pointers, int32_t *const *labels, double *const *centroids, int32_t max_centroids, int32_t *n_centroids,
                        fa_offline_cluster_info *infos, int32_t *statuses) {
    if (!ctx || count < 0 || (count > 0 && (!embeddings || !n || !labels || !n_centroids || !config))) return FA_INVALID_ARGUMENT;
    if (count == 0) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    return guarded(ctx, [&]() -> fa_status {
        // device-resident inputs were produced on the caller's stream; the recordings are prepared on the workers' streams
        if (device_pointers) FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<ClusterJob> jobs;
        jobs.reserve(static_cast<size_t>(count));   // the jobs own device buffers: they must never be copied after prepare()
        std::vector<fa_status> st(static_cast<size_t>(count), FA_SUCCESS);
        for (int32_t r = 0; r < count; ++r) {
            jobs.push_back(ClusterJob{ctx, embeddings[r], n[r], d, rho ? rho[r] : nullptr, rho_dim, chunk_indices ? chunk_indices[r] : nullptr, phi, config, device_pointers,
                                      labels[r], centroids ? centroids[r] : nullptr, max_centroids, &n_centroids[r], infos ? &infos[r] : nullptr});
            st[r] = jobs.back().check_args();
        }
        // Everything except the linkage is, per recording, a chain of small kernels, copies and host decisions: several recordings run
        // it side by side, each worker thread on its own stream of the same device.
        const int workers = std::max(1, std::min<int>(count, 8));
        std::vector<fa_ctx *> wctx(static_cast<size_t>(workers), nullptr);
        wctx[0] = ctx;
        for (int t = 1; t < workers; ++t) {   // worker contexts live with the caller's context (round 4): their streams and buffer caches are reused by the next call
            fa_ctx *&wk = ctx->workers[t - 1];
            if (!wk && fa_ctx_create(ctx->device, nullptr, &wk) != FA_SUCCESS) wk = nullptr;
            if (wk) { wk->ws_limit = ctx->ws_limit; wk->ws_cap = ctx->ws_cap; wk->last_error.clear(); }
            wctx[t] = wk;
        }
        std::vector<std::string> werr(static_cast<size_t>(workers));
        auto on_workers = [&](auto &&phase) {   // phase(job index) -> fa_status, for every recording that is still healthy
            auto work = [&](const int t, const int stride_from) {
                fa_ctx *c = wctx[t];
                fa::DeviceGuard g(c->device);
                for (int32_t r = stride_from; r < count; r += workers) {
                    if (st[r] != FA_SUCCESS) continue;
                    jobs[r].ctx = c;
                    try { st[r] = phase(r); }
                    catch (const std::bad_alloc &) { st[r] = FA_ALLOCATION_FAILURE; }
                    catch (...) { st[r] = FA_UNKNOWN_ERROR; }
                    if (st[r] != FA_SUCCESS && werr[t].empty()) werr[t] = c->last_error;
                    jobs[r].ctx = ctx;
                }
                (void)hipStreamSynchronize(c->stream);
            };
            std::vector<std::thread> th;
            std::vector<char> started(static_cast<size_t>(workers), 0);
            th.reserve(static_cast<size_t>(workers));
            for (int t = 1; t < workers; ++t) if (wctx[t]) started[static_cast<size_t>(t)] = fa::start_thread(th, [&work, t]() { work(t, t); }) ? 1 : 0;
            work(0, 0);
            for (auto &x : th) x.join();
            for (int t = 1; t < workers; ++t) {
                if (!wctx[t]) work(0, t);                               // a worker without a stream of its own: the caller's context takes its share
                else if (!started[static_cast<size_t>(t)]) work(t, t);   // no host thread to be had: the calling thread runs that worker's share on the worker's stream
            }
        };
        on_workers([&](const int32_t r) { return jobs[r].prepare(); });
        // the merge chains of all recordings advance together (one launch = one round of every unfinished recording)
        std::vector<int32_t> who;
        std::vector<const double *> din;
        std::vector<double *> dz;
        std::vector<size_t> rows;
        for (int32_t r = 0; r < count; ++r)
            if (st[r] == FA_SUCCESS && jobs[r].nt >= 2) { who.push_back(r); din.push_back(jobs[r].b_norm.as<double>()); dz.push_back(jobs[r].b_z.as<double>()); rows.push_back(static_cast<size_t>(jobs[r].nt)); }
        std::vector<fa_status> ahc_st(who.size(), FA_SUCCESS);
        std::vector<fa_ahc_stats> ahc_stats(who.size());
        if (!who.empty())
            (void)fa::ahc_run_device_batch(ctx, static_cast<int>(who.size()), din.data(), rows.data(), static_cast<size_t>(d), dz.data(), config->ahc_mode, ahc_stats.data(), ahc_st.data());
        std::vector<fa_status> per_job_ahc(static_cast<size_t>(count), FA_SUCCESS);
        for (size_t j = 0; j < who.size(); ++j) { per_job_ahc[who[j]] = ahc_st[j]; jobs[who[j]].ahc_stats = ahc_stats[j]; }
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        on_workers([&](const int32_t r) { return jobs[r].finish(per_job_ahc[r]); });
        for (int t = 1; t < workers; ++t) if (!werr[t].empty() && ctx->last_error.empty()) ctx->last_error = werr[t];
        fa_status first = FA_SUCCESS;
        for (int32_t r = 0; r < count; ++r) {
            if (statuses) statuses[r] = st[r];
            if (first == FA_SUCCESS && st[r] != FA_SUCCESS) first = st[r];
        }
        return first;
    });
}
}  // namespace

extern "C" {

fa_status fa_offline_cluster_batch(fa_ctx *ctx, int32_t count, const float *const *embeddings, const int64_t *n, int32_t d, const double *const *rho,
                                   int32_t rho_dim, const int32_t *const *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                                   int32_t *const *labels, double *const *centroids, int32_t max_centroids, int32_t *n_centroids,
                                   fa_offline_cluster_info *infos, int32_t *statuses) {
    return cluster_batch(ctx, count, embeddings, n, d, rho, rho_dim, chunk_indices, phi, config, 0, labels, centroids, max_centroids, n_centroids, infos, statuses);
}

fa_status fa_offline_cluster_batch_dev(fa_ctx *ctx, int32_t count, const float *const *d_embeddings, const int64_t *n, int32_t d, const double *const *d_rho,
                                       int32_t rho_dim, const int32_t *const *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                                       int32_t *const *labels, double *const *centroids, int32_t max_centroids, int32_t *n_centroids,
                                       fa_offline_cluster_info *infos, int32_t *statuses) {
    return cluster_batch(ctx, count, d_embeddings, n, d, d_rho, rho_dim, chunk_indices, phi, config, 1, labels, centroids, max_centroids, n_centroids, infos, statuses);
}

}  // extern "C"
