/*
 * fluidaudio_hip.h — C ABI of libfluidaudio_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the host arithmetic of FluidInference/FluidAudio
 * (paths below are relative to the reference's Sources/ directory).
 * Plain pointers and sizes only; no exception crosses this boundary; every entry
 * returns an fa_status whose numbering equals fastcluster_wrapper_status
 * (FastClusterWrapper/include/FastClusterWrapper.h:11-19).
 *
 * Naming: entries ending in _dev take DEVICE pointers and enqueue work on the
 * context's stream without synchronising; the un-suffixed entries take HOST
 * pointers, copy in/out and return when the result is in the caller's buffer
 * (the calling convention of the Swift seams they replace).
 */
#ifndef FLUIDAUDIO_HIP_H
#define FLUIDAUDIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FA_SUCCESS = 0,
    FA_INVALID_ARGUMENT = 1,
    FA_INDEX_OVERFLOW = 2,
    FA_OUTPUT_TOO_SMALL = 3,
    FA_ALLOCATION_FAILURE = 4,
    FA_RUNTIME_ERROR = 5, /* HIP errors, NaN distances, unsupported-on-device conditions */
    FA_UNKNOWN_ERROR = 255
} fa_status;

/* ------------------------------------------------------------------ context -------- */
/* One context = one device + one stream + cached workspaces.  A context is NOT
 * thread-safe; use one per host thread (the reference's AudioMelSpectrogram is likewise
 * one-instance-per-thread, FluidAudio/Shared/AudioMelSpectrogram.swift:48-57). */
typedef struct fa_ctx fa_ctx;

/* stream: a hipStream_t to enqueue on, or NULL to let the context create its own. */
fa_status fa_ctx_create(int device, void *stream, fa_ctx **out);
void fa_ctx_destroy(fa_ctx *ctx);
fa_status fa_ctx_synchronize(fa_ctx *ctx);
/* The hipStream_t every _dev entry of this context enqueues on (for events / stream-ordered callers). */
void *fa_ctx_stream(const fa_ctx *ctx);
/* Page-locked host memory (hipHostMalloc).  Optional: every host-pointer entry accepts ordinary memory; buffers from
 * fa_host_alloc are moved by DMA at the full PCIe rate and let fa_mel_batch overlap its uploads with its downloads. */
void *fa_host_alloc(size_t bytes);
void fa_host_free(void *p);
/* Workspace policy.  A context keeps its linkage workspace (N^2 * 8 B: 15 GB at 43 200 rows, 20 GB at 50 000) and its scratch buffer
 * between calls, because the first call at a new size pays 0.4 - 2.5 s of hipMalloc.
 *   fa_ctx_set_workspace_limit : a cached linkage workspace larger than `bytes` is released when the call that used it returns
 *                                (0 = never keep one).  Default: keep (or the value of FLUIDAUDIO_HIP_WORKSPACE_LIMIT).
 *   fa_ctx_set_workspace_cap   : a linkage call never takes more than `bytes` of workspace.  A problem whose N x N matrix exceeds the cap
 *                                (or HBM, or 196 608 points) runs in the matrix-free mode — O(N d) memory like the reference, the same
 *                                dendrogram, ~5x slower per merge; when even that exceeds the cap the call fails with FA_ALLOCATION_FAILURE
 *                                (the reference's status for std::bad_alloc, FastClusterWrapper.cpp:236-238; AHCClustering degrades to
 *                                singletons).  Default: no cap — hipMalloc decides.
 *   fa_ctx_trim                : releases everything cached now.
 *   fa_ctx_workspace_bytes     : bytes cached right now.
 *   fa_ctx_reserve             : takes the workspace of `recordings` linkage problems of up to n_max points x d dimensions NOW (one
 *                                recording: fa_ahc_linkage / fa_offline_cluster; several: the batched entries, whose problems share one
 *                                allocation), so that a server pays the hipMalloc (0.3 - 6 s for 15 GB, depending on the box) at start-up
 *                                and not inside its first request.  Subject to the cap; kept between calls like any workspace within the limit.
 * Independently of these, a context whose workspace allocation fails first releases the idle caches of the OTHER contexts on the
 * same device and retries, so a pool of contexts on one GPU re-allocates under pressure instead of failing. */
fa_status fa_ctx_set_workspace_limit(fa_ctx *ctx, size_t bytes);
fa_status fa_ctx_set_workspace_cap(fa_ctx *ctx, size_t bytes);
fa_status fa_ctx_trim(fa_ctx *ctx);
size_t fa_ctx_workspace_bytes(const fa_ctx *ctx);
fa_status fa_ctx_reserve(fa_ctx *ctx, size_t n_max, size_t d, int32_t recordings);
/* Last error text recorded on this context ("" if none). */
const char *fa_ctx_last_error(const fa_ctx *ctx);
/* Library build identification, e.g. "fluidaudio_hip 0.1 gfx950". */
const char *fa_version(void);

/* Environment.  The library reads the process environment ONCE (at the first context or switch lookup) and never again:
 *   FLUIDAUDIO_HIP_DEVICES / FLUIDAUDIO_HIP_DEVICE   device set behind the context-free drop-in symbol (fa_pool, below)
 *   FLUIDAUDIO_HIP_WORKSPACE_LIMIT                   default of fa_ctx_set_workspace_limit, bytes
 *   FLUIDAUDIO_HIP_DEBUG_HOOKS=1                     lets fa_debug_inject_fault / fa_debug_set_switch act (tests); without it both are inert
 * ROUTE switches — each forces one of the routes the dispatch chooses between by problem shape, so that every route can be tested on any
 * input; results are identical on every route (that is what the tests check), only the speed differs:
 *   FA_AHC_CPT=1|2|4, FA_AHC_UNI_CPT=1|2|4   slots per thread of the round kernel (single problem / uniform batch)
 *   FA_AHC_NO_SINGLE_BLOCK                   problems of <= 512 points through the multi-block chain
 *   FA_AHC_NO_UNIFORM, FA_AHC_IN_FLIGHT, FA_AHC_UNI_GROUPS=1..4, FA_AHC_UNI_WAVES=6|8   how a batch of problems is laid over launches / streams
 *   FA_AHC_RO_NO_MATRIX                      the reference-order run without the N x N filter matrix (O(N d) memory, like the reference)
 *   FA_AHC_RO_NO_HANDOVER                    AUTO's tie route stays in reference order to the last row (no hand-over to the rounds once the ties have stopped)
 *   FA_AHC_DEBUG                             one line of statistics per linkage call on stderr
 *   FA_MEL_GENERIC, FA_MEL_SLICE_MB=n        the generic mel kernel; slice size of host-pointer batches
 *   FA_VBX_NO_TILED                          the untiled VBx iteration
 *   FA_RESAMPLE_SIMPLE, _NO_DECIM, _NO_DECIM_TILES, _NO_ROWS, _NO_WIDE, FA_RESAMPLE_WIDE=rows:waves (16:8, 16:10, 32:8, 32:10)   polyphase kernel family
 * Switches that only select kernels / parameters for A/B measurements (fa_common.h, FA_SWITCHES' AB list) exist in builds made with
 * -DFA_AB_SWITCHES only.
 * fa_debug_set_switch(name, value): value NULL = unset.  Changes what the NEXT calls see (process-wide; not for use while calls are in flight
 * on other threads).  RUNTIME_ERROR without FLUIDAUDIO_HIP_DEBUG_HOOKS=1, INVALID_ARGUMENT for an unknown name. */
fa_status fa_debug_set_switch(const char *name, const char *value);
int32_t fa_debug_hooks_enabled(void);
/* Test hook: the next `count` passes through `site` fail the way the real failure would (count 0 disarms).  Process-wide; one relaxed
 * atomic load on the paths that carry a site.  Inert unless the process was started with FLUIDAUDIO_HIP_DEBUG_HOOKS=1.  Used by the
 * fault-injection tests of the degrade contracts:
 *   FA_FAULT_VBX            fa::vbx_run_dev returns RUNTIME_ERROR  -> VBxClustering.refine's catch block (VBxClustering.swift:136-141)
 *   FA_FAULT_THREAD_START   no host thread to be had (std::system_error) -> the share runs on the calling thread
 *   FA_FAULT_DEVBUF_MALLOC  the first hipMalloc of a cached-buffer request fails -> idle caches of the device released, retried
 *   FA_FAULT_WS_MALLOC      the first hipMalloc of a linkage workspace fails     -> the same
 *   FA_FAULT_AHC            the linkage of fa_offline_cluster fails -> singletons (AHCClustering.swift:52-55) */
enum { FA_FAULT_VBX = 0, FA_FAULT_THREAD_START = 1, FA_FAULT_DEVBUF_MALLOC = 2, FA_FAULT_WS_MALLOC = 3, FA_FAULT_AHC = 4, FA_FAULT_SITES = 5 };
void fa_debug_inject_fault(int32_t site, int32_t count);
/* Measurement support.  fa_ctx_set_timing(1): entries that support it (fa_ctc_beam_search_batch_dev) bracket the DEVICE work of a call —
 * behind its allocations — with two events on the context's stream; fa_ctx_last_device_ms returns that time (< 0: none recorded), so a
 * caller can tell kernel time from host-side allocation time.  fa_debug_sclk_mhz: the shader clock right now (one wavefront counts
 * s_memtime cycles over spin_us microseconds of the constant 100 MHz s_memrealtime counter). */
fa_status fa_ctx_set_timing(fa_ctx *ctx, int32_t enable);
double fa_ctx_last_device_ms(const fa_ctx *ctx);
fa_status fa_debug_sclk_mhz(fa_ctx *ctx, int32_t spin_us, double *mhz);

/* ------------------------------------------------------------------ mel ------------- */
/* Replaces AudioMelSpectrogram (FluidAudio/Shared/AudioMelSpectrogram.swift):
 *   ctor parameters :59-70, computeFlat :185-292, computeFlatTransposed :325-456,
 *   compute (legacy) :132-178. */
enum { FA_MEL_FLOOR_ADDITIVE = 0, FA_MEL_FLOOR_CLAMPED = 1 };              /* LogFloorMode :24-27 */
enum { FA_MEL_PAD_CENTER = 0, FA_MEL_PAD_PREPADDED = 1, FA_MEL_PAD_LEGACY = 2 }; /* PaddingMode :19-22; LEGACY = compute() */
enum { FA_MEL_LAYOUT_MEL_MAJOR = 0,   /* [n_mels, frames]  computeFlat  (mel[m*stride + t]) */
       FA_MEL_LAYOUT_FRAME_MAJOR = 1  /* [frames, n_mels]  computeFlatTransposed (mel[t*n_mels + m]) */ };

typedef struct {
    int32_t sample_rate;     /* 16000 */
    int32_t n_mels;          /* 128   (1..256) */
    int32_t n_fft;           /* 512   (any power of two 64..2048: LS-EEND uses nextPow2(winLength), LSEENDTypes.swift:55-57;
                                       the batched NeMo configuration n_fft = 512 takes the tuned kernels) */
    int32_t hop;             /* 160 */
    int32_t win;             /* 400   (<= n_fft) */
    float preemph;           /* 0.97 */
    int32_t pad_to;          /* 0 (treated as 1, :72) */
    float log_floor;         /* 2^-24 */
    int32_t floor_mode;      /* FA_MEL_FLOOR_* */
    int32_t window_periodic; /* 0 symmetric / 1 periodic (:553-562) */
    int32_t padding_mode;    /* FA_MEL_PAD_* */
    int32_t layout;          /* FA_MEL_LAYOUT_* */
    /* Extensions beyond AudioMelSpectrogram (fa_mel_default_config sets the values that reproduce it): the torchaudio-
     * flavoured front end of LuxTtsMelExtractor.extract (FluidAudio/TTS/LuxTts/LuxTtsMelExtractor.swift:52-132), the
     * reference's only mel path with a golden vector in its tests. */
    float power;             /* 2 = power spectrum (:459-481; 0 is read as 2), 1 = magnitude (LuxTts :90-96) */
    int32_t center_pad;      /* FA_MEL_CENTER_* ; only meaningful with FA_MEL_PAD_CENTER */
    int32_t mel_scale;       /* FA_MEL_SCALE_* */
    int32_t tail_mode;       /* FA_MEL_TAIL_* */
    const float *filterbank; /* optional HOST table [n_mels][n_fft/2 + 1] that replaces the built-in bank; NULL = mel_scale */
} fa_mel_config;
enum { FA_MEL_CENTER_ZERO = 0,     /* zero padding of n_fft/2 (:206-217) */
       FA_MEL_CENTER_REFLECT = 1   /* torch pad_mode="reflect" (LuxTts :58-66) */ };
enum { FA_MEL_SCALE_SLANEY = 0,    /* Slaney scale, area-normalised triangles (:564-642) */
       FA_MEL_SCALE_HTK_NONORM = 1 /* torchaudio melscale_fbanks(norm: nil, mel_scale: "htk") (LuxTts :160-189) */ };
enum { FA_MEL_TAIL_ZERO = 0,       /* frames past the signal see zeros (truncated windows, :412) */
       FA_MEL_TAIL_REPLICATE = 1   /* frames >= the signal's own frame count repeat its last frame (lhotse alignment, LuxTts :124-128) */ };

void fa_mel_default_config(fa_mel_config *cfg);
/* Frame count T the reference would emit for n_samples (:192-197, :335-347, :133); 0 when its guard fires. */
int32_t fa_mel_num_frames(const fa_mel_config *cfg, int64_t n_samples);
/* ceil(T / pad_to) * pad_to (:204, :354). */
int32_t fa_mel_padded_frames(const fa_mel_config *cfg, int32_t frames);

/* A plan fixes the batch geometry (utterance offsets), the tables (Hann window, Slaney
 * filterbank in sparse form) and the launch grid; executing it is a pure kernel launch. */
typedef struct fa_mel_plan fa_mel_plan;

/* offsets: HOST array of B+1 sample offsets into the packed pcm buffer (utterance b is
 *          pcm[offsets[b] .. offsets[b+1])).
 * expected_frames: HOST array of B frame-count overrides (expectedFrameCount, :329,:347)
 *          or NULL.
 * frame_stride: allocated frames per utterance in the output (>= padded frames of every
 *          utterance); 0 = use the largest padded frame count in the batch.
 * Output addressing: utterance b starts at mel + b * fa_mel_plan_utt_stride(plan);
 *   MEL_MAJOR  : mel[b][m][t] at m*frame_stride + t     FRAME_MAJOR: mel[b][t][m] at t*n_mels + m
 * Frames t >= T(b) inside the allocation are written as 0 (padValue, :39). */
fa_status fa_mel_plan_create(fa_ctx *ctx, const fa_mel_config *cfg, const int64_t *offsets, int32_t batch,
                             const int32_t *expected_frames, int32_t frame_stride, fa_mel_plan **out);
void fa_mel_plan_destroy(fa_mel_plan *plan);
int64_t fa_mel_plan_utt_stride(const fa_mel_plan *plan);   /* floats per utterance in the output */
int32_t fa_mel_plan_frame_stride(const fa_mel_plan *plan);
int64_t fa_mel_plan_total_frames(const fa_mel_plan *plan); /* sum of T(b) */

/* d_pcm: device float[offsets[B]]; d_last_samples: device float[B] (lastAudioSample, :186) or NULL = 0;
 * d_mel: device float[B * utt_stride]; d_mel_lengths: device int32[B] (melLength) or NULL. */
fa_status fa_mel_execute_dev(fa_mel_plan *plan, const float *d_pcm, const float *d_last_samples,
                             float *d_mel, int32_t *d_mel_lengths);

/* Host-buffer convenience (the shape of a loop of computeFlat calls): copies pcm in, runs
 * the plan, copies mel (+lengths) out, synchronises.  When pcm AND mel are page-locked (fa_host_alloc) the batch is
 * processed in ~64 MB slices whose uploads overlap the downloads of the slices before them (both PCIe directions busy). */
fa_status fa_mel_batch(fa_ctx *ctx, const fa_mel_config *cfg, const float *pcm, const int64_t *offsets,
                       int32_t batch, const float *last_samples, const int32_t *expected_frames,
                       int32_t frame_stride, float *mel, int32_t *mel_lengths);

/* NeMo per_feature normalisation as applied by UnifiedMelExtractor.normalizePerFeature
 * (FluidAudio/ASR/Parakeet/Unified/UnifiedMelExtractor.swift:91-113), in place on a MEL_MAJOR tensor
 * d_mel[batch][n_mels][frame_stride]: per (utterance, mel) row, over the first valid_frames[b] frames, subtract the mean
 * and divide by (unbiased std + 1e-5); frames valid..frames-1 become 0.  valid_frames[b] = min(validCount / hop, T) is
 * the caller's (:66). */
fa_status fa_mel_normalize_per_feature_dev(fa_ctx *ctx, float *d_mel, int32_t batch, int32_t n_mels, int32_t frame_stride,
                                           int32_t frames, const int32_t *d_valid_frames);

/* Host copies of the tables (createHannWindow :553-562, createMelFilterbank :564-642). */
fa_status fa_mel_hann_window(const fa_mel_config *cfg, float *out /* win */);
fa_status fa_mel_filterbank(const fa_mel_config *cfg, float *out /* n_mels * (n_fft/2+1) */);

/* ------------------------------------------------------------------ argmax / CTC ---- */
/* Replaces LogitsArgmax.argmaxPerFrame (FluidAudio/ASR/Shared/LogitsArgmax.swift:16-55) and the
 * greedy collapse of ctcGreedyDecode (FluidAudio/ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:15-36,
 * 45-70; FluidAudio/ASR/SenseVoice/SenseVoiceManager.swift:119-126). */
enum { FA_DTYPE_F32 = 0, FA_DTYPE_F16 = 1 };

/* logits: batch matrices, matrix b at element offset b*matrix_stride, row t at t*row_stride,
 *         `vocab` valid elements per row (padding columns are never read).
 * valid_frames: int32[batch] (frames to decode per matrix, <= frames) or NULL = all.
 * frame_ids (optional): int32[batch * frames] per-frame argmax (argmaxPerFrame output).
 * token_ids: int32[batch * frames] collapsed ids (first token_lens[b] valid per matrix).
 * Ties -> lowest index; NaN never wins; an all-NaN/-inf row yields 0. */
fa_status fa_ctc_greedy_batch_dev(fa_ctx *ctx, const void *d_logits, int32_t dtype, int32_t batch, int32_t frames,
                                  int32_t vocab, int64_t row_stride, int64_t matrix_stride,
                                  const int32_t *d_valid_frames, int32_t blank_id, int32_t *d_frame_ids,
                                  int32_t *d_token_ids, int32_t *d_token_lens);
fa_status fa_ctc_greedy_batch(fa_ctx *ctx, const void *logits, int32_t dtype, int32_t batch, int32_t frames,
                              int32_t vocab, int64_t row_stride, int64_t matrix_stride,
                              const int32_t *valid_frames, int32_t blank_id, int32_t *frame_ids,
                              int32_t *token_ids, int32_t *token_lens);

/* ctcGreedyDecode(logProbs: [[Float]], ...) — the array-of-arrays overload (CtcDecoder.swift:15-36), whose semantics differ from the
 * [1, T, V] overload above (:45-70) in three ways, all mirrored here:
 *   - the scan is seeded with frame[0] (:25), so a frame whose element 0 is NaN decodes to index 0;
 *   - every frame has its own length (:26): row r is values[row_offsets[r] .. row_offsets[r+1]);
 *   - an empty frame is skipped before `prev` is updated (:23): `a, [], a` collapses to ONE a.
 * utt_rows: int64[batch+1] row ranges of the utterances (NULL with batch == 1: all rows are one utterance).
 * frame_ids (optional): int32[total_rows], -1 for an empty frame.  token_ids: int32[total_rows]; utterance u writes its
 * token_lens[u] ids from token_ids[utt_rows[u]].  fp32 only ([[Float]]). */
fa_status fa_ctc_greedy_rows_dev(fa_ctx *ctx, const float *d_values, const int64_t *d_row_offsets, int64_t total_rows,
                                 const int64_t *d_utt_rows, int32_t batch, int32_t blank_id, int32_t *d_frame_ids,
                                 int32_t *d_token_ids, int32_t *d_token_lens);
fa_status fa_ctc_greedy_rows(fa_ctx *ctx, const float *values, const int64_t *row_offsets, int64_t total_rows,
                             const int64_t *utt_rows, int32_t batch, int32_t blank_id, int32_t *frame_ids,
                             int32_t *token_ids, int32_t *token_lens);

/* Per-frame log-softmax with temperature and blank bias: CtcKeywordSpotter.makeLogProbs / logSoftmax
 * (FluidAudio/ASR/Parakeet/SlidingWindow/CustomVocabulary/WordSpotting/CtcKeywordSpotter+Inference.swift:350-431).
 * DEVICE pointers; logits addressed like fa_ctc_greedy_batch_dev; d_log_probs: float[batch][frames][vocab] contiguous.
 * blank_bias is subtracted from column blank_id when != 0 (:397-399); temperature divides the logits when != 1 (:412). */
fa_status fa_ctc_log_softmax_batch_dev(fa_ctx *ctx, const void *d_logits, int32_t dtype, int32_t batch, int32_t frames,
                                       int32_t vocab, int64_t row_stride, int64_t matrix_stride, float temperature,
                                       float blank_bias, int32_t blank_id, float *d_log_probs);

/* ------------------------------------------------------------------ TDT ------------- */
/* Control flow of TdtDecoderV3.decodeWithTimings (FluidAudio/ASR/Parakeet/SlidingWindow/TDT/Decoder/TdtDecoderV3.swift:103-607)
 * and its helpers (TdtFrameNavigation.swift:20-105, TdtDurationMapping.swift:17-31, TdtConfig.swift:13-26).  The decoder
 * LSTM and joint network are CoreML models outside the reference tree: token parity is UNPINNED; the device entry
 * replays the loop over tables of the joint's decisions indexed by (decoder steps taken u, encoder frame t). */
typedef struct {
    int32_t blank_id;                /* 8192 */
    int32_t max_symbols_per_step;    /* 10 */
    int32_t max_tokens_per_chunk;    /* 150 */
    int32_t consecutive_blank_limit; /* 5 */
    int32_t n_duration_bins;         /* 5 (<= 8) */
    int32_t duration_bins[8];        /* 0 1 2 3 4 */
} fa_tdt_config;
void fa_tdt_default_config(fa_tdt_config *cfg);
/* calculateInitialTimeIndices (TdtFrameNavigation.swift:20-49); has_time_jump == 0 <=> timeJump == nil */
int32_t fa_tdt_initial_time_index(int32_t has_time_jump, int32_t time_jump, int32_t context_frame_adjustment);
/* initializeNavigationState (:59-78) */
void fa_tdt_navigation_state(int32_t time_indices, int32_t encoder_sequence_length, int32_t actual_audio_frames,
                             int32_t *effective_length, int32_t *safe_time_indices, int32_t *last_timestep, int32_t *active);
/* calculateFinalTimeJump (:91-105); *has_value == 0 <=> nil (last chunk) */
int32_t fa_tdt_final_time_jump(int32_t current_time_indices, int32_t effective_length, int32_t is_last_chunk, int32_t *has_value);
/* mapDurationBin (TdtDurationMapping.swift:17-22): RUNTIME_ERROR when out of range */
fa_status fa_tdt_map_duration_bin(const fa_tdt_config *cfg, int32_t bin_index, int32_t *duration);
/* clampProbability (:28-31) */
float fa_tdt_clamp_probability(float value);
/* Batched greedy walk, one chunk per table set.  DEVICE pointers.  d_tok/d_bin/d_prob: [batch][U][T] joint decisions;
 * per chunk: d_enc_len (encoderSequenceLength), d_audio_frames (NULL = enc_len), d_t0 (initial time index, NULL = 0),
 * d_is_last (NULL = 0), d_global_offset (NULL = 0), d_emit_after (emitTokensAfterGlobalFrame, NULL or < 0 = nil).
 * Outputs per chunk: up to max_out (token, global timestamp, duration, confidence), their count, the final time index
 * (INT32_MIN when the reference returns before touching timeJump), decoder steps taken, and a status (RUNTIME_ERROR for
 * a duration bin out of range, OUTPUT_TOO_SMALL when U or max_out is exhausted). */
fa_status fa_tdt_greedy_tables_dev(fa_ctx *ctx, const fa_tdt_config *cfg, const int32_t *d_tok, const int32_t *d_bin,
                                   const float *d_prob, int32_t batch, int32_t U, int32_t T, const int32_t *d_enc_len,
                                   const int32_t *d_audio_frames, const int32_t *d_t0, const int32_t *d_is_last,
                                   const int32_t *d_global_offset, const int32_t *d_emit_after, int32_t max_out,
                                   int32_t *d_out_tok, int32_t *d_out_time, int32_t *d_out_dur, float *d_out_conf,
                                   int32_t *d_out_count, int32_t *d_final_time, int32_t *d_final_u, int32_t *d_status);

/* The same walk with the joint decisions computed on the fly from joint LOGITS: d_logits[batch][U][T][row_stride] (fp32 or
 * fp16), token logits in [0, vocab_with_blank), the n_duration_bins duration logits right behind them.  Per visited (u, t):
 * token = first-index argmax of the token logits (the tie / NaN rule of LogitsArgmax.argmaxPerFrame), probability = its
 * softmax probability, duration bin = first-index argmax of the duration logits — what the reference's JointDecision model
 * returns (TdtModelInference.swift:84-188).  One workgroup per chunk reads only the rows on the greedy path.  The joint /
 * decoder networks are not in the reference tree: token parity UNPINNED. */
fa_status fa_tdt_greedy_logits_dev(fa_ctx *ctx, const fa_tdt_config *cfg, const void *d_logits, int32_t dtype, int32_t batch,
                                   int32_t U, int32_t T, int32_t vocab_with_blank, int64_t row_stride, const int32_t *d_enc_len,
                                   const int32_t *d_audio_frames, const int32_t *d_t0, const int32_t *d_is_last,
                                   const int32_t *d_global_offset, const int32_t *d_emit_after, int32_t max_out,
                                   int32_t *d_out_tok, int32_t *d_out_time, int32_t *d_out_dur, float *d_out_conf,
                                   int32_t *d_out_count, int32_t *d_final_time, int32_t *d_final_u, int32_t *d_status);

/* ------------------------------------------------------------------ AHC ------------- */
/* Exact signature + status contract of the reference FFI
 * (FastClusterWrapper/include/FastClusterWrapper.h:35-41, FastClusterWrapper.cpp:196-244):
 * HOST pointers, row-major data, SciPy-format dendrogram rows in merge order.  Declared with
 * the reference's own names in include/FastClusterWrapper.h (included here):
 *   fastcluster_wrapper_status fastcluster_compute_centroid_linkage(const double *data,
 *       size_t pointCount, size_t dimension, double *dendrogramOut, size_t dendrogramLength); */
#include "FastClusterWrapper.h"

typedef struct {
    int64_t merges;        /* N-1 */
    int64_t rounds;        /* select/apply kernel pairs executed */
    int64_t rescans;       /* lazy row re-scans; in a reference-order run through the matrix filter: + the rows whose candidates were too many for
                            * one wavefront and were scanned again with exact sums */
    int64_t exact_fallback;/* 1 if the Lance-Williams filter hit an ambiguity and the run switched to exact rows */
    double init_ms, merge_ms, total_ms; /* device time (hipEvent) */
    int64_t windows;       /* rounds in which several pairs fell inside the rounding bound and were re-evaluated exactly */
    int64_t reference_order; /* 1 if the run met an EXACT tie at the minimum (or was asked to) and was computed in the reference's own
                              * selection order (binary heap + index-ordered scans, fastcluster_internal.hpp:778-935,1625-1800):
                              * row for row the reference's output on tied input; `rounds` then counts its scans */
    int64_t handed_over_at;  /* AUTO's tie route: rows computed in reference order before the rest of the problem went back to the filter-based rounds
                              * (ties at distance 0 only — duplicates — that had stopped); 0: not handed over, -1: handed over, met a tie, recomputed */
} fa_ahc_stats;

enum { FA_AHC_MODE_AUTO = 0,   /* Lance-Williams filter + exact re-verification; an exact tie at the minimum (or a window overflowing
                                * with near-ties) re-runs the problem in reference order: the dendrogram is the reference's row for row */
       FA_AHC_MODE_EXACT = 1,  /* every matrix entry is the reference's sequential fp64 sum; ties in (value, row, column) order */
       FA_AHC_MODE_REFERENCE_ORDER = 2 };/* the reference's selection order from the start (what AUTO falls back to): ~12 us per merge at 43 200 points with the
                                         * distance matrix as the filter of its scans, 25-28 us matrix-free (no N x N workspace to be had, or FA_AHC_RO_NO_MATRIX set) */

/* Same computation with an explicit context; data/dendrogram are HOST pointers unless
 * device_pointers != 0.  stats may be NULL. */
fa_status fa_ahc_linkage(fa_ctx *ctx, const double *data, size_t n, size_t d, double *dendrogram,
                         size_t dendrogram_len, int32_t mode, int32_t device_pointers, fa_ahc_stats *stats);

/* `count` independent problems (one per recording) of dimension d in ONE call: the serial merge chains advance together —
 * one launch is one round of every unfinished problem — so many medium-sized recordings fill the machine that a single chain
 * leaves idle.  data / dendrograms: HOST arrays of `count` pointers (to host buffers, or to device buffers when
 * device_pointers != 0); n: HOST array of row counts; statuses (nullable): per-problem status with the contract of the
 * reference symbol; stats (nullable): `count` entries (init_ms / merge_ms are those of the whole batch).  Returns the
 * first per-problem failure.  Every dendrogram equals the one fa_ahc_linkage produces for that problem alone. */
fa_status fa_ahc_linkage_batch(fa_ctx *ctx, int32_t count, const double *const *data, const size_t *n, size_t d,
                               double *const *dendrograms, int32_t mode, int32_t device_pointers, fa_ahc_stats *stats,
                               int32_t *statuses);

/* Nearest other point of rows [row0, row1) among all n rows of x (n x d, row-major fp64): mins[i - row0] = the reference's
 * distance (sequential sum of squared differences, FastClusterWrapper.cpp:45-52; the SQUARED distance like the linkage uses
 * before its final sqrt), args[i - row0] = its index (lowest on ties).  The start-up table of the linkage
 * (fastcluster_internal.hpp:1653-1678) in a form that shards by rows across GPUs: every rank holds all of x (all-gather),
 * computes its slab, and the (min, arg) pairs are gathered (fluidaudio_amd/sharding.py::row_minima_sharded). */
fa_status fa_ahc_row_minima(fa_ctx *ctx, const double *x, size_t n, size_t d, size_t row0, size_t row1, double *mins, int32_t *args,
                            int32_t device_pointers);

/* AHCClustering.cluster (FluidAudio/Diarizer/Offline/Clustering/AHCClustering.swift:20-67): L2-normalise
 * (:70-105), linkage, threshold clamp (:112-121), top-down cut (:124-197), relabel (:200-210).
 * x: HOST double[n*d]; labels: HOST int32[n].  On linkage failure labels = 0..n-1 (:52-55) and
 * the failing status is returned. */
fa_status fa_ahc_cluster(fa_ctx *ctx, const double *x, size_t n, size_t d, double threshold, int32_t mode,
                         int32_t *labels, fa_ahc_stats *stats);
/* The cut alone (:124-210) on a host dendrogram. */
fa_status fa_ahc_cut(const double *dendrogram, size_t n, double threshold, int32_t *labels);

/* ------------------------------------------------------------------ VBx ------------- */
/* VBxClustering.refine / runVBx (FluidAudio/Diarizer/Offline/Clustering/VBxClustering.swift:41-165,167-664).
 * HOST pointers.  rho: double[T*D]; initial: int32[T] AHC labels; phi: double[D].
 * gamma: double[T*S] out, pi: double[S] out, hard: int32[T] out (argmax, first max :144-146),
 * elbos: double[max_iter] out, n_iters out.  S = number of distinct labels in `initial`
 * (returned through n_speakers; gamma/pi must be sized for it — query with fa_vbx_speaker_count). */
int32_t fa_vbx_speaker_count(const int32_t *initial, int64_t T);
fa_status fa_vbx_refine(fa_ctx *ctx, const double *rho, int64_t T, int32_t D, const int32_t *initial,
                        const double *phi, double Fa, double Fb, int32_t max_iter, double epsilon,
                        double *gamma, double *pi, int32_t *hard, double *elbos, int32_t *n_iters,
                        int32_t *n_speakers);

/* The same iteration loop (VBxClustering.swift:301-661) sharded over the frame axis, one fa_vbx_shard per device.
 * Everything that crosses frames is one record per slice of the frame axis (fa_vbx_shard_slices() = 64 slices of ceil(T / 64) frames:
 * [S][D + 1] doubles = sum_t gamma[t][s] (rho[t][:], 1), then the slice's sum of the per-frame log-likelihoods); a device holds
 * 64 / world consecutive slices = the frames fa_vbx_shard_range() names (world must divide 64).  Protocol, identical on every rank:
 *     begin(chunk)  ->  all-gather the chunks in rank order into `full`
 *     repeat up to max_iter times { iterate(full, chunk) -> all-gather -> finish_iteration(full, &elbo); stop when |elbo - previous| < epsilon (:653-659) }
 *     result(...)
 * Every rank evaluates speaker statistics, pi and the ELBO from the same gathered records: same decisions everywhere, no other
 * collective.  The records of a slice are computed exactly as fa_vbx_refine computes them on one device, so the sharded run equals
 * fa_vbx_refine bit for bit at every world size.  rho_local / labels_local: HOST, the frames of fa_vbx_shard_range(); phi HOST [D];
 * chunk / full: DEVICE double[fa_vbx_shard_chunk_doubles(S, D, world)] / [... (S, D, 1)] (the caller's collective library moves them;
 * both calls that write a chunk return with it complete); S = number of distinct labels of the WHOLE problem. */
typedef struct fa_vbx_shard fa_vbx_shard;
int32_t fa_vbx_shard_slices(void);
void fa_vbx_shard_range(int64_t T_total, int32_t rank, int32_t world, int64_t *t_lo, int64_t *t_hi);
int64_t fa_vbx_shard_chunk_doubles(int32_t S, int32_t D, int32_t world);
fa_status fa_vbx_shard_create(fa_ctx *ctx, const double *rho_local, int64_t T_total, int32_t D, const int32_t *labels_local, int32_t S,
                              const double *phi, double Fa, double Fb, int32_t rank, int32_t world, fa_vbx_shard **out);
void fa_vbx_shard_destroy(fa_vbx_shard *shard);
void fa_vbx_shard_frames(const fa_vbx_shard *shard, int64_t *t_lo, int64_t *t_hi);
fa_status fa_vbx_shard_begin(fa_vbx_shard *shard, double *d_chunk);
fa_status fa_vbx_shard_iterate(fa_vbx_shard *shard, const double *d_full, double *d_chunk);
fa_status fa_vbx_shard_finish_iteration(fa_vbx_shard *shard, const double *d_full, double *elbo);
/* HOST outputs (each may be NULL): gamma_local double[frames held * S], pi double[S], hard_local int32[frames held]. */
fa_status fa_vbx_shard_result(fa_vbx_shard *shard, double *gamma_local, double *pi, int32_t *hard_local);

/* ------------------------------------------------------------------ post-VBx -------- */
/* OfflineDiarizerManager.computeCentroids (FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:613-691) and
 * assignEmbeddings (:789-822).  HOST pointers, fp64.
 * emb: double[n*d]; gamma: double[n*S]; pi: double[S].  Speakers with pi > 1e-7 are kept: map[s] = row of `centroids`
 * (or -1), *n_centroids = number kept; centroids must hold S*d doubles.  Summation order = the reference's. */
fa_status fa_vbx_weighted_centroids(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *gamma,
                                    const double *pi, int32_t S, double *centroids, int32_t *map, int32_t *n_centroids);
/* out[i] = argmax_k cosine(emb_i, centroid_k), first maximum; K == 0 -> all 0 (:795-797). */
fa_status fa_assign_cosine(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *centroids, int32_t K,
                           int32_t *out);
/* centroidScores (:789-798): scores[i*K + k] = cosine(emb_i, centroid_k). */
fa_status fa_centroid_scores(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, const double *centroids, int32_t K,
                             double *scores);
/* ConstrainedClusterAssignment.assign (FluidAudio/Diarizer/Offline/Clustering/ConstrainedClusterAssignment.swift:20-42):
 * per chunk, HungarianAssignment.maxScoreAssignment (FluidAudio/Diarizer/HungarianAssignment.swift:67-97) — distinct
 * local speakers of a chunk get distinct clusters, maximising the total score; -2 = slot dropped (more speakers than
 * clusters).  scores: double[n*K]; chunk_indices: int32[n]; out: int32[n].  At most 256 rows per chunk / clusters. */
fa_status fa_constrained_assign(fa_ctx *ctx, const double *scores, int64_t n, int32_t K, const int32_t *chunk_indices,
                                int32_t *out);

/* ------------------------------------------------------------------ the clustering stage, composed ---- */
/* OfflineDiarizerManager.cluster (FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:270-375) on precomputed
 * embeddings as ONE call whose intermediates never leave HBM: selectTrainingEmbeddings (:591-611) -> AHCClustering.cluster
 * (:301-306) -> VBxClustering.refineWithConstraints (:308-333) -> computeCentroids / computeCentroidsFromClusters (:613-740)
 * -> constrained per-chunk assignment or cosine argmax of every embedding (:345-375, :789-822).  Config defaults =
 * OfflineDiarizerTypes.swift:155-163,189-192. */
typedef struct {
    double clustering_threshold;     /* 0.6 */
    double warm_start_fa;            /* 0.07 */
    double warm_start_fb;            /* 0.8 */
    int32_t max_vbx_iterations;      /* 20 */
    double convergence_tolerance;    /* 1e-4 */
    int32_t constrained_assignment;  /* 1 */
    int64_t num_speakers;            /* -1 = nil */
    int64_t min_speakers;            /* -1 = nil */
    int64_t max_speakers;            /* -1 = nil */
    int32_t ahc_mode;                /* FA_AHC_MODE_AUTO */
} fa_offline_cluster_config;
typedef struct {
    int64_t training_rows;           /* rows that passed selectTrainingEmbeddings */
    int32_t initial_clusters;        /* distinct AHC labels */
    int32_t vbx_iterations;
    int32_t was_adjusted;            /* the K-Means fallback replaced the VBx posteriors (VBxOutput.wasAdjusted) */
    int32_t constrained;             /* the constrained per-chunk assignment was used */
    int32_t vbx_degraded;            /* VBx failed: gamma = one-hot AHC labels, pi = 1/S, no ELBOs, the stage went on (VBxClustering.swift:136-141) */
    int32_t ahc_degraded;            /* the linkage failed: every training row its own cluster (AHCClustering.swift:52-55) */
    double inputs_s, ahc_s, vbx_s, assign_s, total_s;   /* host wall-clock per stage (copies included) */
    fa_ahc_stats ahc;
} fa_offline_cluster_info;
void fa_offline_cluster_default_config(fa_offline_cluster_config *cfg);
/* embeddings: float[n*d] (the 256-d speaker embeddings, widened to fp64 on the device like :286); rho: double[n*rho_dim]
 * PLDA features (rho_dim == 0: no VBx, centroids come from the AHC labels); chunk_indices: HOST int32[n] (needed when
 * constrained_assignment != 0); phi: HOST double[rho_dim].  embeddings / rho are HOST pointers unless device_pointers != 0.
 * labels: HOST int32[n] out (-2 = slot dropped by the constrained assignment); centroids: HOST double[max_centroids*d] out or
 * NULL; *n_centroids out; info may be NULL.  n == 0 -> INVALID_ARGUMENT (the reference throws noSpeechDetected, :281-283). */
fa_status fa_offline_cluster(fa_ctx *ctx, const float *embeddings, int64_t n, int32_t d, const double *rho, int32_t rho_dim,
                             const int32_t *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                             int32_t device_pointers, int32_t *labels, double *centroids, int32_t max_centroids,
                             int32_t *n_centroids, fa_offline_cluster_info *info);
/* The same call with copies of the stage's intermediates (verification at full size: bench.py and the 8 h digest test compare them
 * with the CPU side): ahc_labels HOST int32[training rows] = AHCClustering.cluster's labels (AHCClustering.swift:20-67), vbx_hard
 * HOST int32[training rows] = argmax of the VBx posteriors (VBxClustering.swift:144-146), elbos HOST double[max_vbx_iterations]
 * (info->vbx_iterations of them are written).  Each may be NULL. */
fa_status fa_offline_cluster_ex(fa_ctx *ctx, const float *embeddings, int64_t n, int32_t d, const double *rho, int32_t rho_dim,
                                const int32_t *chunk_indices, const double *phi, const fa_offline_cluster_config *config,
                                int32_t device_pointers, int32_t *labels, double *centroids, int32_t max_centroids,
                                int32_t *n_centroids, fa_offline_cluster_info *info, int32_t *ahc_labels, int32_t *vbx_hard,
                                double *elbos);
/* `count` recordings through the same stage in ONE call: inputs and training rows of every recording are prepared, the merge
 * chains of all of them advance together (fa_ahc_linkage_batch's round launches: a single chain leaves most of the machine
 * idle), then every recording is cut / refined / assigned.  HOST pointers; all recordings share d, rho_dim, phi and config.
 * embeddings / rho / chunk_indices / labels / centroids: arrays of `count` pointers (rho and centroids may be NULL as in the
 * single call); n, n_centroids, infos (nullable), statuses (nullable): `count` entries.  Per recording the results equal
 * fa_offline_cluster's bit for bit; a failing recording does not stop the others; the first failure is returned.
 * infos[r].*_s of a batch are wall-clock marks of the shared phases, not per-recording costs. */
fa_status fa_offline_cluster_batch(fa_ctx *ctx, int32_t count, const float *const *embeddings, const int64_t *n, int32_t d,
                                   const double *const *rho, int32_t rho_dim, const int32_t *const *chunk_indices, const double *phi,
                                   const fa_offline_cluster_config *config, int32_t *const *labels, double *const *centroids,
                                   int32_t max_centroids, int32_t *n_centroids, fa_offline_cluster_info *infos, int32_t *statuses);
/* The same with the recordings' inputs RESIDENT on the context's device (what device_pointers = 1 is to fa_offline_cluster): d_embeddings[r] /
 * d_rho[r] are DEVICE pointers (the arrays of pointers themselves, n, chunk_indices[r], phi, labels[r], centroids[r] stay on the host).  The call
 * waits for the context's stream first (the inputs' producer), nothing is uploaded.  Results equal fa_offline_cluster_batch's bit for bit. */
fa_status fa_offline_cluster_batch_dev(fa_ctx *ctx, int32_t count, const float *const *d_embeddings, const int64_t *n, int32_t d,
                                       const double *const *d_rho, int32_t rho_dim, const int32_t *const *chunk_indices, const double *phi,
                                       const fa_offline_cluster_config *config, int32_t *const *labels, double *const *centroids,
                                       int32_t max_centroids, int32_t *n_centroids, fa_offline_cluster_info *infos, int32_t *statuses);


/* ------------------------------------------------ speaker-count constraints + K-Means fallback ------ */
/* KMeansClustering.SeededRNG.next (FluidAudio/Diarizer/Offline/Clustering/KMeansClustering.swift:212-223) and the Swift
 * standard library's RandomNumberGenerator.next(upperBound:) over it (Lemire's method; the draw behind shuffle(using:) and
 * randomElement(using:)).  Pure host functions. */
uint64_t fa_seeded_rng_next(uint64_t *state);
uint64_t fa_seeded_rng_below(uint64_t *state, uint64_t upper_bound);
/* KMeansClustering.clusterWithCentroids (:39-91): unit-normalise, centroids = first k of a seeded shuffle, Lloyd iterations
 * until the assignment repeats or max_iterations; empty clusters re-seeded from a random embedding.  HOST pointers:
 * emb double[n*d] -> labels int32[n], centroids double[min(k,n)*d] (nullable), *out_k = centroid rows written (0 for the
 * degenerate returns: d == 0 or num_clusters <= 0 -> all labels 0; n <= k -> labels 0..n-1 and the raw embeddings). */
fa_status fa_kmeans_cluster(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, int32_t num_clusters, int32_t max_iterations,
                            uint64_t seed, int32_t *labels, double *centroids, int32_t *out_k, int32_t *out_iterations);
/* KMeansClustering.clusterWithCentroidsNInit (:99-129): seeds base_seed + 0..n_init-1, lowest inertia wins (first on ties).
 * All runs advance together on the device.  best_run / inertias[n_init] nullable. */
fa_status fa_kmeans_cluster_ninit(fa_ctx *ctx, const double *emb, int64_t n, int32_t d, int32_t num_clusters, int32_t max_iterations,
                                  int32_t n_init, uint64_t base_seed, int32_t *labels, double *centroids, int32_t *out_k,
                                  int32_t *best_run, double *inertias);
/* SpeakerCountConstraints.resolve (FluidAudio/Diarizer/Offline/Clustering/SpeakerCountConstraints.swift:25-62).  A null
 * pointer is Swift's nil.  out = { numSpeakers or -1 for nil, minSpeakers, maxSpeakers }. */
void fa_speaker_constraints_resolve(int64_t num_embeddings, const int64_t *num_speakers, const int64_t *min_speakers,
                                    const int64_t *max_speakers, int64_t out[3]);

/* ------------------------------------------------------------------ CTC beam search + ARPA LM ------ */
/* ARPALanguageModel (FluidAudio/ASR/Parakeet/SlidingWindow/CTC/ARPALanguageModel.swift:16-104): unigrams and bigrams of a
 * plain-text ARPA file (tab-separated fields, log10 -> natural log in float); higher orders ignored, malformed lines
 * skipped.  Words are keyed by a 64-bit polynomial hash of their bytes plus their length.  Parsing and fa_arpa_score are
 * host code (ctx may be NULL); the tables move to the device on the first search. */
typedef struct fa_arpa_lm fa_arpa_lm;
fa_status fa_arpa_parse(fa_ctx *ctx, const char *text, int64_t len, fa_arpa_lm **out);
void fa_arpa_destroy(fa_arpa_lm *lm);
int64_t fa_arpa_unigram_count(const fa_arpa_lm *lm);          /* lm.unigrams.count */
int64_t fa_arpa_bigram_context_count(const fa_arpa_lm *lm);   /* lm.bigrams.count (context words) */
/* ARPALanguageModel.score(word:prev:) (:98-103); prev may be NULL (nil).  Pure host function on the same tables. */
fa_status fa_arpa_score(const fa_arpa_lm *lm, const char *word, const char *prev, float *out);

/* The token vocabulary ([Int: String]) as the device needs it for word tracking: SentencePiece word-boundary flag and the
 * hash pair of the piece.  ids[i] -> pieces[i] (UTF-8); ids not listed decode to "". */
typedef struct fa_ctc_vocab fa_ctc_vocab;
fa_status fa_ctc_vocab_create(fa_ctx *ctx, const int32_t *ids, const char *const *pieces, int32_t n, int32_t vocab_size,
                              fa_ctc_vocab **out);
void fa_ctc_vocab_destroy(fa_ctc_vocab *vocab);

/* ctcBeamSearch (FluidAudio/ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:118-241) for a batch of [frames, vocab] float
 * log-probability matrices: one workgroup per utterance.  lm / vocabulary may be NULL (no rescoring).  beam_width <= 128,
 * token_candidates <= 64.  tokens: int32[batch][frames] (best prefix, lens[b] entries used), scores: total of the winner
 * (nullable).  The string form is decodeCtcTokenIds over tokens.  DEVICE pointers, synchronous. */
fa_status fa_ctc_beam_search_batch_dev(fa_ctx *ctx, const float *d_log_probs, int32_t batch, int32_t frames, int32_t vocab,
                                       int64_t row_stride, int64_t matrix_stride, const int32_t *d_valid_frames,
                                       const fa_ctc_vocab *vocabulary, fa_arpa_lm *lm, int32_t beam_width, float lm_weight,
                                       float word_bonus, int32_t blank_id, int32_t token_candidates, int32_t *d_tokens,
                                       int32_t *d_lens, float *d_scores);
/* Same with HOST pointers and contiguous matrices. */
/* What a search of these shapes launches: out = { trie slots per utterance, utterances per launch (the ~2 GiB arena cap), launches,
 * extension keys per thread of the ctc_beam_kernel instance (8 / 20 / 32) }.  Host only. */
fa_status fa_ctc_beam_plan(int32_t batch, int32_t frames, int32_t vocab, int32_t beam_width, int32_t blank_id, int32_t token_candidates,
                           int64_t out[4]);
fa_status fa_ctc_beam_search_batch(fa_ctx *ctx, const float *log_probs, int32_t batch, int32_t frames, int32_t vocab,
                                   const int32_t *valid_frames, const fa_ctc_vocab *vocabulary, fa_arpa_lm *lm,
                                   int32_t beam_width, float lm_weight, float word_bonus, int32_t blank_id,
                                   int32_t token_candidates, int32_t *tokens, int32_t *lens, float *scores);

/* ------------------------------------------------------------------ wire formats ------ */
/* AudioWAV.data (FluidAudio/Shared/AudioConverter.swift:474-532): float samples -> peak normalisation (normalize != 0 and
 * max |x| > 0) -> clamp to [-1, 1] -> Int16(x * 32767) -> 16-bit PCM mono RIFF/WAVE (44-byte header + 2 n bytes).  HOST
 * pointers; the sample pass runs on the device and is bit-identical to the reference's Float arithmetic. */
int64_t fa_wav_pcm16_size(int64_t n_samples);
fa_status fa_wav_encode_pcm16(fa_ctx *ctx, const float *samples, int64_t n, double sample_rate, int32_t normalize, uint8_t *out,
                              int64_t out_capacity, int64_t *out_len);
/* Extension (the reference reads audio through AVFoundation): RIFF/WAVE reader for 16-bit PCM and 32-bit float data,
 * interleaved float output; out may be NULL to query frames / channels / sample_rate.  Pure host function. */
fa_status fa_wav_decode(const uint8_t *data, int64_t len, float *out, int64_t out_capacity, int64_t *frames, int32_t *channels,
                        int32_t *sample_rate);

typedef struct fa_rttm_segment {   /* TimedSpeakerSegment as the RTTM loaders fill it */
    float start_seconds, end_seconds, quality;
    char speaker_id[64];
} fa_rttm_segment;
/* RTTMParser.loadSegments (FluidAudioCLI/Utils/RTTMParser.swift:22-63; strict = 1: blank and '#' lines skipped, any other
 * malformed line is an error reported in bad_line, result sorted by start time) or the benchmark loader
 * (FluidAudioCLI/Commands/SortformerBenchmark.swift:681-731; strict = 0: malformed lines skipped, file order kept).
 * Pure host function; *count receives the number of segments even when out is NULL / too small. */
fa_status fa_rttm_parse(const char *text, int64_t len, int32_t strict, fa_rttm_segment *out, int64_t out_capacity, int64_t *count,
                        char *bad_line, int64_t bad_line_capacity);
/* Extension: one "SPEAKER <file> 1 <start> <duration> <NA> <NA> <speaker> <NA> <NA>" line per segment; returns the text
 * length and writes it (NUL-terminated) when out_capacity is larger; -1 when the text could not be built (host allocation failed). */
int64_t fa_rttm_format(const fa_rttm_segment *segs, int64_t n, const char *file_id, char *out, int64_t out_capacity);

typedef struct fa_export_embedding {   /* TimedEmbedding fields of the export payload */
    int32_t chunk_index, speaker_index, start_frame, end_frame;
    double start_time, end_time;
} fa_export_embedding;
/* OfflineDiarizerManager.exportEmbeddings (FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:913-955): JSON
 * array of {chunkIndex, speakerIndex, startFrame, endFrame, startTime, endTime, embedding256, rho128, cluster}; cluster = -1
 * past the end of assignments.  Numbers are printed in their shortest round-trip form.  Returns the text length (-1: host allocation failed). */
int64_t fa_export_embeddings_json(const fa_export_embedding *items, int64_t n, const float *embedding256, int32_t emb_dim,
                                  const double *rho128, int32_t rho_dim, const int32_t *assignments, int64_t n_assignments,
                                  char *out, int64_t out_capacity);

/* ------------------------------------------------------------------ resampling ------ */
/* AudioConverter.linearResample (FluidAudio/Shared/AudioConverter.swift:388-442): planar float[channels][frames] ->
 * mono mix (weight 1/channels) -> linear interpolation to out_rate.  HOST pointers.  Bit-exact restatement. */
int64_t fa_resample_linear_frames(int64_t frames, double in_rate, double out_rate);
fa_status fa_resample_linear(fa_ctx *ctx, const float *planar, int32_t channels, int64_t frames, double in_rate,
                             double out_rate, float *out, int64_t out_capacity, int64_t *out_frames);
/* EXTENSION (parity unpinned: the reference delegates to Apple's closed-source AVAudioConverter, :299-370):
 * rational polyphase FIR resampler, Kaiser(5.0) windowed sinc, half length 10*max(up,down), output length
 * ceil(frames*up/down) — the specification of scipy.signal.resample_poly.  up/down are reduced by their gcd. */
int64_t fa_resample_poly_frames(int64_t frames, int32_t up, int32_t down);
fa_status fa_resample_poly_taps(int32_t up, int32_t down, float *taps, int64_t capacity, int64_t *n_taps, int64_t *pre_remove);
fa_status fa_resample_poly(fa_ctx *ctx, const float *x, int64_t frames, int32_t up, int32_t down, float *out,
                           int64_t out_capacity, int64_t *out_frames);
/* The same on device-resident buffers, enqueued on the context's stream (no synchronisation): d_x float[frames] ->
 * d_y float[fa_resample_poly_frames(frames, up, down)]. */
fa_status fa_resample_poly_dev(fa_ctx *ctx, const float *d_x, int64_t frames, int32_t up, int32_t down, float *d_y,
                               int64_t out_capacity, int64_t *out_frames);

/* ------------------------------------------------------------------ device set ------ */
/* Multi-GPU for a single-process host (the reference has no multi-device notion; SURVEY §8e: utterances, logit matrices and
 * recordings are independent units, nothing is exchanged).  A pool holds one context per listed device; listing a device
 * twice gives two contexts (two streams) on it.  devices == NULL / n_devices == 0: every visible device.
 *   * fa_pool_acquire blocks until a context is free and hands it to the calling thread; fa_pool_release returns it.
 *     The context-free drop-in symbol fastcluster_compute_centroid_linkage does exactly this on a default pool built from
 *     FLUIDAUDIO_HIP_DEVICES="0,1,.." (default: all devices; FLUIDAUDIO_HIP_DEVICE=<one id> is still honoured), so concurrent
 *     AHCClustering.cluster calls (FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:270) run on different GPUs.
 *   * the _sharded / _many entries split ONE host-pointer call across all contexts of the pool (contiguous utterance ranges
 *     balanced by samples; matrices by count; recordings dealt longest-first), one host thread per context, results written
 *     straight into the caller's buffers with the geometry of the unsharded entry; they return the first failure. */
typedef struct fa_pool fa_pool;
fa_status fa_device_count(int32_t *count);
fa_status fa_pool_create(const int32_t *devices, int32_t n_devices, fa_pool **out);
void fa_pool_destroy(fa_pool *pool);
int32_t fa_pool_size(const fa_pool *pool);
fa_ctx *fa_pool_context(fa_pool *pool, int32_t index);   /* the pool keeps ownership */
int32_t fa_ctx_device(const fa_ctx *ctx);
fa_status fa_pool_acquire(fa_pool *pool, fa_ctx **ctx);
void fa_pool_release(fa_pool *pool, fa_ctx *ctx);
fa_status fa_mel_batch_sharded(fa_pool *pool, const fa_mel_config *cfg, const float *pcm, const int64_t *offsets, int32_t batch,
                               const float *last_samples, const int32_t *expected_frames, int32_t frame_stride, float *mel,
                               int32_t *mel_lengths);
fa_status fa_ctc_greedy_batch_sharded(fa_pool *pool, const void *logits, int32_t dtype, int32_t batch, int32_t frames, int32_t vocab,
                                      int64_t row_stride, int64_t matrix_stride, const int32_t *valid_frames, int32_t blank_id,
                                      int32_t *frame_ids, int32_t *token_ids, int32_t *token_lens);
/* recordings across devices, and on each device advanced together by fa_ahc_linkage_batch; HOST pointers */
fa_status fa_ahc_linkage_many(fa_pool *pool, int32_t count, const double *const *data, const size_t *n, size_t d,
                              double *const *dendrograms, int32_t mode, fa_ahc_stats *stats, int32_t *statuses);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDAUDIO_HIP_H */
