/*
 * fa_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the FluidAudio host arithmetic that the MI355X library
 * replaces.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path (fluidaudio_amd/, libfluidaudio_hip.so)
 * never links, imports or calls anything declared here.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/Sources/FluidAudio unless noted).
 *
 * Pinning status:
 *   - AHC linkage: the reference's own C++ is built unmodified into
 *     oracle/_ref/libfastcluster_ref.so (see oracle/Makefile); fa_oracle_linkage_naive
 *     below is a second, independent restatement of the same greedy semantics.
 *   - AHC pre/post, argmax, CTC collapse: pinned by the reference's XCTest
 *     known-answer cases (ported in tests/).
 *   - generic STFT->mel machinery: pinned by the LuxTTS golden fixture.
 *   - AudioMelSpectrogram NeMo-flavoured config, VBx: PARITY UNPINNED beyond this
 *     restatement (no golden values exist in the reference; Accelerate is closed).
 */
#ifndef FA_ORACLE_H
#define FA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- mel (Shared/AudioMelSpectrogram.swift) ---------------------------------- */

typedef struct {
    int sample_rate;     /* 16000 */
    int n_mels;          /* 128 */
    int n_fft;           /* 512 (power of two) */
    int hop;             /* 160 */
    int win;             /* 400 */
    float preemph;       /* 0.97 */
    int pad_to;          /* 0 -> treated as 1 (:72) */
    float log_floor;     /* 2^-24 */
    int floor_clamped;   /* 0 = additive log(v+floor), 1 = clamped log(max(v,floor)) (:542-549) */
    int window_periodic; /* 0 = symmetric (NeMo), 1 = periodic (:553-562) */
} fa_oracle_mel_config;

void fa_oracle_mel_default_config(fa_oracle_mel_config *cfg);

/* createHannWindow (:553-562). out[win]. */
void fa_oracle_hann(int win, int periodic, float *out);
/* createMelFilterbank (:564-642). out[n_mels * (n_fft/2+1)] row-major. */
void fa_oracle_slaney_filterbank(int n_fft, int n_mels, int sample_rate, float *out);

/* Frame-count helpers (:192-204, :335-354).  Return T (0 when the guard fires). */
int fa_oracle_mel_frames_center(const fa_oracle_mel_config *cfg, long n_samples);
int fa_oracle_mel_frames_prepadded(const fa_oracle_mel_config *cfg, long n_samples);
int fa_oracle_mel_padded_frames(const fa_oracle_mel_config *cfg, int frames);

/* computeFlat (:185-292): output [n_mels, Tpad] (mel[m*Tpad+t]).
 * out must hold n_mels * max(Tpad,1) floats.  Returns 0 on success. */
int fa_oracle_mel_flat(const fa_oracle_mel_config *cfg, const float *audio, long n_samples,
                       float last_sample, float *out, int *mel_length, int *num_frames);

/* computeFlatTransposed (:325-456): output [Tpad, n_mels] (mel[t*n_mels+m]).
 * prepadded: 0 = .center, 1 = .prePadded.  expected_frames < 0 means nil. */
int fa_oracle_mel_flat_transposed(const fa_oracle_mel_config *cfg, const float *audio, long n_samples,
                                  float last_sample, int prepadded, int expected_frames,
                                  float *out, int *mel_length, int *num_frames);

/* compute (:132-178) legacy: no preemph, no centre pad, window at frame[0..win).
 * output [n_mels, T].  Returns T (<=0: empty). */
int fa_oracle_mel_legacy(const fa_oracle_mel_config *cfg, const float *audio, long n_samples, float *out);

/* Generic STFT->mel used only to pin the shared machinery against the LuxTTS golden
 * fixture (TTS/LuxTts/LuxTtsMelExtractor.swift:52-132): reflect pad n_fft/2, window of
 * n_fft samples, magnitude (power=1) or power (=2), dense fb, log(max(v,floor)).
 * out [frames, n_mels]. */
int fa_oracle_logmel_generic(const float *audio, long n, int n_fft, int hop, const float *window,
                             const float *fb, int n_mels, int power, float floor_v, int frames,
                             float *out);

/* fp32 radix-2 FFT used by the restatement (stand-in for vDSP_DFT_zop, :459-471). */
void fa_oracle_fft_f32(int n, float *re, float *im);

/* ---- argmax + CTC greedy (ASR/Shared/LogitsArgmax.swift:16-55,
 *      ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:45-70) ---------------------- */

/* argmaxPerFrame: rows t<frames, V elements at p + t*row_stride.  is_f16: logits are
 * IEEE binary16 widened to fp32 first (:33-52). */
void fa_oracle_argmax_rows(const void *logits, int is_f16, long frames, long vocab, long row_stride,
                           int32_t *ids);
/* CTC collapse (:53-68): returns number of emitted ids. */
long fa_oracle_ctc_collapse(const int32_t *frame_ids, long frames, int32_t blank_id, int32_t *out);
/* argmax + collapse in one call. */
long fa_oracle_ctc_greedy(const void *logits, int is_f16, long frames, long vocab, long row_stride,
                          int32_t blank_id, int32_t *out);
/* ctcGreedyDecode(logProbs: [[Float]]) CtcDecoder.swift:15-36: frame[0] seed, per-frame lengths, empty frames skipped */
long fa_oracle_ctc_greedy_rows(const float *values, const int64_t *row_offsets, long rows, int32_t blank_id,
                               int32_t *frame_ids, int32_t *out);

/* ---- AHC pre/post (Diarizer/Offline/Clustering/AHCClustering.swift) ------------- */

/* normalizeFeatures (:70-105). */
void fa_oracle_ahc_normalize(const double *x, long n, long d, double *out);
/* clampDistanceThreshold (:112-121). */
double fa_oracle_ahc_clamp_threshold(double thr);
/* assignmentsFromDendrogram + remapClusterIds (:124-210). */
void fa_oracle_ahc_cut(const double *dendrogram, long n, double threshold, int32_t *labels);

/* Type of the reference C ABI (Sources/FastClusterWrapper/include/FastClusterWrapper.h:35-41). */
typedef int (*fa_oracle_linkage_fn)(const double *, size_t, size_t, double *, size_t);
/* AHCClustering.cluster (:20-67) with the linkage supplied by the caller
 * (oracle/_ref build of the reference, or the library under test). */
int fa_oracle_ahc_cluster(fa_oracle_linkage_fn linkage, const double *x, long n, long d,
                          double threshold, int32_t *labels);

/* Independent restatement of the linkage semantics
 * (Sources/FastClusterWrapper/fastcluster_internal.hpp:1625-1800 +
 *  FastClusterWrapper.cpp:45-52,89-100,128-130,169-192): at every step merge the
 * globally closest pair of active centroids, distances = sequential fp64
 * sum (x-y)^2, ties -> lowest (larger index, smaller index).  O(N^2 d + N^2) memory-free
 * version, for small N only.  Returns a FastClusterWrapper status code. */
int fa_oracle_linkage_naive(const double *data, size_t n, size_t d, double *z, size_t zlen);

/* ---- VBx (Diarizer/Offline/Clustering/VBxClustering.swift:41-165,167-664) -------- */

/* runVBx.  gamma_io [T,S] in: initial gamma (one-hot) / out: final gamma.  pi_out[S].
 * elbos_out[max_iter].  Returns number of iterations run. */
int fa_oracle_vbx_run(const double *features, long T, long D, const double *phi,
                      double *gamma_io, long S, int max_iter, double epsilon,
                      double Fa, double Fb, double init_smoothing,
                      double *pi_out, double *elbos_out);
/* refine (:41-165): builds the one-hot gamma from initial cluster labels, runs VBx,
 * returns hard assignments (first max).  S = number of distinct labels. */
int fa_oracle_vbx_refine(const double *rho, long T, long D, const int32_t *initial, const double *phi,
                         int max_iter, double epsilon, double Fa, double Fb,
                         double *gamma_out, double *pi_out, int32_t *hard_out, double *elbos_out,
                         long *S_out);

/* ---- post-VBx (Diarizer/Offline/Core/OfflineDiarizerManager.swift:613-691,789-822) -- */

/* gamma-weighted centroids for speakers with pi > 1e-7.  centroids_out [K,d]; map_out[S]
 * = centroid row or -1.  Returns K. */
long fa_oracle_weighted_centroids(const double *emb, long n, long d, const double *gamma, const double *pi,
                                  long S, double *centroids_out, int32_t *map_out);
/* cosine argmax assignment (first max). */
void fa_oracle_assign_cosine(const double *emb, long n, long d, const double *centroids, long K,
                             int32_t *out);

/* AudioConverter.linearResample (FluidAudio/Shared/AudioConverter.swift:388-442) */
long fa_oracle_resample_linear_frames(long frames, double in_rate, double out_rate);
long fa_oracle_resample_linear(const float *planar, int channels, long frames, double in_rate, double out_rate, float *out);

/* UnifiedMelExtractor.normalizePerFeature (FluidAudio/ASR/Parakeet/Unified/UnifiedMelExtractor.swift:91-113) */
void fa_oracle_normalize_per_feature(float *x_time_major, int n_mels, int frames, int valid_frames);

/* HungarianAssignment (FluidAudio/Diarizer/HungarianAssignment.swift:8-97), ConstrainedClusterAssignment
 * (FluidAudio/Diarizer/Offline/Clustering/ConstrainedClusterAssignment.swift:20-42), centroidScores
 * (FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:789-798) */
void fa_oracle_hungarian_solve(const long long *cost, int n, int *assign);
void fa_oracle_max_score_assignment(const double *scores, int rows, int cols, int *assign);
void fa_oracle_constrained_assign(const double *scores, long n, int K, const int32_t *chunk, int32_t *out);
void fa_oracle_centroid_scores(const double *emb, long n, long d, const double *centroids, long K, double *scores);

/* TDT greedy control flow (FluidAudio/ASR/Parakeet/SlidingWindow/TDT/Decoder/TdtDecoderV3.swift:103-607,
 * TdtFrameNavigation.swift:20-105, TdtDurationMapping.swift:17-31); joint decisions served from [U][T] tables */
int fa_oracle_tdt_initial_time_index(int has_time_jump, int time_jump, int context_frame_adjustment);
float fa_oracle_tdt_clamp_probability(float v);
long fa_oracle_tdt_last_joint_calls(void);   /* joint evaluations of this thread's last fa_oracle_tdt_greedy call */
int fa_oracle_tdt_greedy(const int32_t *tok, const int32_t *bin, const float *prob, int U, int T, int enc_len, int audio_frames,
                         int t0, int is_last, int global_offset, int emit_after, int blank_id, int max_symbols, int max_tokens,
                         int blank_limit, const int *bins, int nbins, int max_out, int32_t *out_tok, int32_t *out_time,
                         int32_t *out_dur, float *out_conf, int *out_count, int *final_time, int *final_u);

/* CtcKeywordSpotter.logSoftmax + blank bias (CtcKeywordSpotter+Inference.swift:397-431) */
void fa_oracle_log_softmax_row(const float *logits, int V, float temperature, float blank_bias, int blank_id, float *out);

/* K-Means fallback (KMeansClustering.swift:39-224) with the Swift-stdlib draws restated; SpeakerCountConstraints.resolve */
uint64_t fa_oracle_seeded_rng_next(uint64_t *state);
uint64_t fa_oracle_rng_upper_bound(uint64_t *state, uint64_t bound);
void fa_oracle_shuffle_indices(uint64_t *state, long n, int64_t *idx);
void fa_oracle_kmeans_normalize(const double *x, long n, long d, double *out);
int fa_oracle_kmeans(const double *emb, long n, long d, long num_clusters, long max_iter, uint64_t seed,
                     int32_t *labels, double *centroids, long *out_k, long *out_iters);
int fa_oracle_kmeans_ninit(const double *emb, long n, long d, long num_clusters, long max_iter, long n_init, uint64_t base_seed,
                           int32_t *labels, double *centroids, long *out_k, long *best_run, double *inertias);
void fa_oracle_speaker_constraints(long num_embeddings, int has_num, long num, int has_min, long mn, int has_max, long mx, long out[3]);

#ifdef __cplusplus
}
#endif
#endif
