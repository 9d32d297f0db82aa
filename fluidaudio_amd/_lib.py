"""ctypes binding of libfluidaudio_hip.so (the C ABI declared in include/fluidaudio_hip.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, the product
path raises.  (The oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FLUIDAUDIO_HIP_LIBRARY") or os.path.join(CSRC, "libfluidaudio_hip.so")   # the override is for kernel experiments (scripts/)

SUCCESS, INVALID_ARGUMENT, INDEX_OVERFLOW, OUTPUT_TOO_SMALL, ALLOCATION_FAILURE, RUNTIME_ERROR, UNKNOWN_ERROR = 0, 1, 2, 3, 4, 5, 255
STATUS_NAMES = {0: "SUCCESS", 1: "INVALID_ARGUMENT", 2: "INDEX_OVERFLOW", 3: "OUTPUT_TOO_SMALL",
                4: "ALLOCATION_FAILURE", 5: "RUNTIME_ERROR", 255: "UNKNOWN_ERROR"}

MEL_FLOOR_ADDITIVE, MEL_FLOOR_CLAMPED = 0, 1
MEL_PAD_CENTER, MEL_PAD_PREPADDED, MEL_PAD_LEGACY = 0, 1, 2
MEL_LAYOUT_MEL_MAJOR, MEL_LAYOUT_FRAME_MAJOR = 0, 1
MEL_CENTER_ZERO, MEL_CENTER_REFLECT = 0, 1
MEL_SCALE_SLANEY, MEL_SCALE_HTK_NONORM = 0, 1
MEL_TAIL_ZERO, MEL_TAIL_REPLICATE = 0, 1
DTYPE_F32, DTYPE_F16 = 0, 1
AHC_MODE_AUTO, AHC_MODE_EXACT, AHC_MODE_REFERENCE_ORDER = 0, 1, 2
FAULT_VBX, FAULT_THREAD_START, FAULT_DEVBUF_MALLOC, FAULT_WS_MALLOC, FAULT_AHC = 0, 1, 2, 3, 4   # fa_debug_inject_fault sites (tests)

# Every symbol include/fluidaudio_hip.h + include/FastClusterWrapper.h declare (checked by tests/test_abi.py).
EXPORTED_SYMBOLS = [
    "fa_version", "fa_debug_inject_fault", "fa_debug_set_switch", "fa_debug_hooks_enabled", "fa_ctx_set_timing", "fa_ctx_last_device_ms", "fa_debug_sclk_mhz", "fa_ctc_beam_plan", "fa_host_alloc", "fa_host_free", "fa_ctx_create", "fa_ctx_destroy", "fa_ctx_synchronize", "fa_ctx_stream", "fa_ctx_last_error",
    "fa_ctx_set_workspace_limit", "fa_ctx_set_workspace_cap", "fa_ctx_trim", "fa_ctx_workspace_bytes", "fa_ctx_reserve",
    "fa_mel_default_config", "fa_mel_num_frames", "fa_mel_padded_frames", "fa_mel_plan_create", "fa_mel_plan_destroy",
    "fa_mel_plan_utt_stride", "fa_mel_plan_frame_stride", "fa_mel_plan_total_frames", "fa_mel_execute_dev",
    "fa_mel_batch", "fa_mel_hann_window", "fa_mel_filterbank", "fa_mel_normalize_per_feature_dev",
    "fa_ctc_greedy_batch_dev", "fa_ctc_greedy_batch", "fa_ctc_greedy_rows_dev", "fa_ctc_greedy_rows", "fa_ctc_log_softmax_batch_dev",
    "fa_tdt_default_config", "fa_tdt_initial_time_index", "fa_tdt_navigation_state", "fa_tdt_final_time_jump",
    "fa_tdt_map_duration_bin", "fa_tdt_clamp_probability", "fa_tdt_greedy_tables_dev", "fa_tdt_greedy_logits_dev",
    "fastcluster_compute_centroid_linkage", "fa_ahc_linkage", "fa_ahc_linkage_batch", "fa_ahc_row_minima", "fa_ahc_cluster", "fa_ahc_cut",
    "fa_vbx_speaker_count", "fa_vbx_refine",
    "fa_vbx_shard_slices", "fa_vbx_shard_range", "fa_vbx_shard_chunk_doubles", "fa_vbx_shard_create", "fa_vbx_shard_destroy",
    "fa_vbx_shard_frames", "fa_vbx_shard_begin", "fa_vbx_shard_iterate", "fa_vbx_shard_finish_iteration", "fa_vbx_shard_result",
    "fa_vbx_weighted_centroids", "fa_assign_cosine", "fa_centroid_scores", "fa_constrained_assign",
    "fa_offline_cluster_default_config", "fa_offline_cluster", "fa_offline_cluster_ex", "fa_offline_cluster_batch", "fa_offline_cluster_batch_dev",
    "fa_arpa_parse", "fa_arpa_destroy", "fa_arpa_unigram_count", "fa_arpa_bigram_context_count", "fa_arpa_score",
    "fa_ctc_vocab_create", "fa_ctc_vocab_destroy", "fa_ctc_beam_search_batch_dev", "fa_ctc_beam_search_batch",
    "fa_wav_pcm16_size", "fa_wav_encode_pcm16", "fa_wav_decode", "fa_rttm_parse", "fa_rttm_format", "fa_export_embeddings_json",
    "fa_seeded_rng_next", "fa_seeded_rng_below", "fa_kmeans_cluster", "fa_kmeans_cluster_ninit", "fa_speaker_constraints_resolve",
    "fa_resample_linear_frames", "fa_resample_linear", "fa_resample_poly_frames", "fa_resample_poly_taps", "fa_resample_poly", "fa_resample_poly_dev",
    "fa_device_count", "fa_pool_create", "fa_pool_destroy", "fa_pool_size", "fa_pool_context", "fa_ctx_device", "fa_pool_acquire", "fa_pool_release",
    "fa_mel_batch_sharded", "fa_ctc_greedy_batch_sharded", "fa_ahc_linkage_many",
]


class FluidAudioHipError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}" + (f" ({detail})" if detail else ""))


class MelConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_mels", C.c_int32), ("n_fft", C.c_int32), ("hop", C.c_int32),
                ("win", C.c_int32), ("preemph", C.c_float), ("pad_to", C.c_int32), ("log_floor", C.c_float),
                ("floor_mode", C.c_int32), ("window_periodic", C.c_int32), ("padding_mode", C.c_int32),
                ("layout", C.c_int32), ("power", C.c_float), ("center_pad", C.c_int32), ("mel_scale", C.c_int32),
                ("tail_mode", C.c_int32), ("filterbank", C.c_void_p)]


class TdtConfig(C.Structure):
    _fields_ = [("blank_id", C.c_int32), ("max_symbols_per_step", C.c_int32), ("max_tokens_per_chunk", C.c_int32),
                ("consecutive_blank_limit", C.c_int32), ("n_duration_bins", C.c_int32), ("duration_bins", C.c_int32 * 8)]


class AhcStats(C.Structure):
    _fields_ = [("merges", C.c_int64), ("rounds", C.c_int64), ("rescans", C.c_int64), ("exact_fallback", C.c_int64),
                ("init_ms", C.c_double), ("merge_ms", C.c_double), ("total_ms", C.c_double), ("windows", C.c_int64), ("reference_order", C.c_int64), ("handed_over_at", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class OfflineClusterConfig(C.Structure):
    _fields_ = [("clustering_threshold", C.c_double), ("warm_start_fa", C.c_double), ("warm_start_fb", C.c_double),
                ("max_vbx_iterations", C.c_int32), ("convergence_tolerance", C.c_double), ("constrained_assignment", C.c_int32),
                ("num_speakers", C.c_int64), ("min_speakers", C.c_int64), ("max_speakers", C.c_int64), ("ahc_mode", C.c_int32)]


class OfflineClusterInfo(C.Structure):
    _fields_ = [("training_rows", C.c_int64), ("initial_clusters", C.c_int32), ("vbx_iterations", C.c_int32),
                ("was_adjusted", C.c_int32), ("constrained", C.c_int32), ("vbx_degraded", C.c_int32), ("ahc_degraded", C.c_int32),
                ("inputs_s", C.c_double), ("ahc_s", C.c_double),
                ("vbx_s", C.c_double), ("assign_s", C.c_double), ("total_s", C.c_double), ("ahc", AhcStats)]


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into csrc/libfluidaudio_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8", "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libfluidaudio_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the product path)")
    try:
        # torch's wheel bundles its own ROCm runtime: when libfluidaudio_hip.so pulls /opt/rocm's libamdhip64 into the process
        # FIRST, a later `import torch` sees no device (measured on the MI355X box).  Loading torch first makes both use one runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t
    L.fa_version.restype = C.c_char_p
    L.fa_debug_inject_fault.argtypes = [i32, i32]
    L.fa_debug_inject_fault.restype = None
    L.fa_debug_set_switch.argtypes = [C.c_char_p, C.c_char_p]
    L.fa_debug_hooks_enabled.restype = i32
    L.fa_ctx_set_timing.argtypes = [vp, i32]
    L.fa_ctx_last_device_ms.argtypes = [vp]
    L.fa_ctx_last_device_ms.restype = f64
    L.fa_debug_sclk_mhz.argtypes = [vp, i32, vp]
    L.fa_ctc_beam_plan.argtypes = [i32, i32, i32, i32, i32, i32, vp]
    L.fa_host_alloc.argtypes = [sz]
    L.fa_host_alloc.restype = vp
    L.fa_host_free.argtypes = [vp]
    L.fa_host_free.restype = None
    L.fa_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.fa_ctx_destroy.argtypes = [vp]
    L.fa_ctx_destroy.restype = None
    L.fa_ctx_synchronize.argtypes = [vp]
    L.fa_ctx_stream.argtypes = [vp]
    L.fa_ctx_stream.restype = vp
    L.fa_ctx_last_error.argtypes = [vp]
    L.fa_ctx_set_workspace_limit.argtypes = [vp, sz]
    L.fa_ctx_set_workspace_cap.argtypes = [vp, sz]
    L.fa_ctx_trim.argtypes = [vp]
    L.fa_ctx_workspace_bytes.argtypes = [vp]
    L.fa_ctx_workspace_bytes.restype = sz
    L.fa_ctx_reserve.argtypes = [vp, sz, sz, i32]
    L.fa_ctx_last_error.restype = C.c_char_p
    L.fa_mel_default_config.argtypes = [C.POINTER(MelConfig)]
    L.fa_mel_default_config.restype = None
    L.fa_mel_num_frames.argtypes = [C.POINTER(MelConfig), i64]
    L.fa_mel_num_frames.restype = i32
    L.fa_mel_padded_frames.argtypes = [C.POINTER(MelConfig), i32]
    L.fa_mel_padded_frames.restype = i32
    L.fa_mel_plan_create.argtypes = [vp, C.POINTER(MelConfig), vp, i32, vp, i32, C.POINTER(vp)]
    L.fa_mel_plan_destroy.argtypes = [vp]
    L.fa_mel_plan_destroy.restype = None
    L.fa_mel_plan_utt_stride.argtypes = [vp]
    L.fa_mel_plan_utt_stride.restype = i64
    L.fa_mel_plan_frame_stride.argtypes = [vp]
    L.fa_mel_plan_frame_stride.restype = i32
    L.fa_mel_plan_total_frames.argtypes = [vp]
    L.fa_mel_plan_total_frames.restype = i64
    L.fa_mel_execute_dev.argtypes = [vp, vp, vp, vp, vp]
    L.fa_mel_batch.argtypes = [vp, C.POINTER(MelConfig), vp, vp, i32, vp, vp, i32, vp, vp]
    L.fa_mel_hann_window.argtypes = [C.POINTER(MelConfig), vp]
    L.fa_mel_normalize_per_feature_dev.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.fa_mel_filterbank.argtypes = [C.POINTER(MelConfig), vp]
    L.fa_ctc_greedy_batch_dev.argtypes = [vp, vp, i32, i32, i32, i32, i64, i64, vp, i32, vp, vp, vp]
    L.fa_ctc_log_softmax_batch_dev.argtypes = [vp, vp, i32, i32, i32, i32, i64, i64, f32, f32, i32, vp]
    L.fa_ctc_greedy_batch.argtypes = [vp, vp, i32, i32, i32, i32, i64, i64, vp, i32, vp, vp, vp]
    L.fa_ctc_greedy_rows_dev.argtypes = [vp, vp, vp, i64, vp, i32, i32, vp, vp, vp]
    L.fa_ctc_greedy_rows.argtypes = [vp, vp, vp, i64, vp, i32, i32, vp, vp, vp]
    L.fa_tdt_default_config.argtypes = [C.POINTER(TdtConfig)]
    L.fa_tdt_default_config.restype = None
    L.fa_tdt_initial_time_index.argtypes = [i32, i32, i32]
    L.fa_tdt_initial_time_index.restype = i32
    L.fa_tdt_navigation_state.argtypes = [i32, i32, i32] + [C.POINTER(i32)] * 4
    L.fa_tdt_navigation_state.restype = None
    L.fa_tdt_final_time_jump.argtypes = [i32, i32, i32, C.POINTER(i32)]
    L.fa_tdt_final_time_jump.restype = i32
    L.fa_tdt_map_duration_bin.argtypes = [C.POINTER(TdtConfig), i32, C.POINTER(i32)]
    L.fa_tdt_clamp_probability.argtypes = [f32]
    L.fa_tdt_clamp_probability.restype = f32
    L.fa_tdt_greedy_tables_dev.argtypes = [vp, C.POINTER(TdtConfig), vp, vp, vp, i32, i32, i32] + [vp] * 6 + [i32] + [vp] * 8
    L.fa_tdt_greedy_logits_dev.argtypes = [vp, C.POINTER(TdtConfig), vp, i32, i32, i32, i32, i32, i64] + [vp] * 6 + [i32] + [vp] * 8
    L.fastcluster_compute_centroid_linkage.argtypes = [vp, sz, sz, vp, sz]
    L.fastcluster_compute_centroid_linkage.restype = C.c_int
    L.fa_ahc_linkage.argtypes = [vp, vp, sz, sz, vp, sz, i32, i32, C.POINTER(AhcStats)]
    L.fa_ahc_linkage_batch.argtypes = [vp, i32, vp, vp, sz, vp, i32, i32, vp, vp]
    L.fa_ahc_row_minima.argtypes = [vp, vp, sz, sz, sz, sz, vp, vp, i32]
    L.fa_ahc_cluster.argtypes = [vp, vp, sz, sz, f64, i32, vp, C.POINTER(AhcStats)]
    L.fa_ahc_cut.argtypes = [vp, sz, f64, vp]
    L.fa_vbx_speaker_count.argtypes = [vp, i64]
    L.fa_vbx_speaker_count.restype = i32
    L.fa_vbx_refine.argtypes = [vp, vp, i64, i32, vp, vp, f64, f64, i32, f64, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.fa_vbx_shard_slices.argtypes = []
    L.fa_vbx_shard_slices.restype = i32
    L.fa_vbx_shard_range.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.fa_vbx_shard_range.restype = None
    L.fa_vbx_shard_chunk_doubles.argtypes = [i32, i32, i32]
    L.fa_vbx_shard_chunk_doubles.restype = i64
    L.fa_vbx_shard_create.argtypes = [vp, vp, i64, i32, vp, i32, vp, f64, f64, i32, i32, C.POINTER(vp)]
    L.fa_vbx_shard_destroy.argtypes = [vp]
    L.fa_vbx_shard_destroy.restype = None
    L.fa_vbx_shard_frames.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.fa_vbx_shard_frames.restype = None
    L.fa_vbx_shard_begin.argtypes = [vp, vp]
    L.fa_vbx_shard_iterate.argtypes = [vp, vp, vp]
    L.fa_vbx_shard_finish_iteration.argtypes = [vp, vp, C.POINTER(f64)]
    L.fa_vbx_shard_result.argtypes = [vp, vp, vp, vp]
    L.fa_vbx_weighted_centroids.argtypes = [vp, vp, i64, i32, vp, vp, i32, vp, vp, C.POINTER(i32)]
    L.fa_assign_cosine.argtypes = [vp, vp, i64, i32, vp, i32, vp]
    L.fa_centroid_scores.argtypes = [vp, vp, i64, i32, vp, i32, vp]
    L.fa_constrained_assign.argtypes = [vp, vp, i64, i32, vp, vp]
    L.fa_offline_cluster_default_config.argtypes = [C.POINTER(OfflineClusterConfig)]
    L.fa_offline_cluster_default_config.restype = None
    L.fa_offline_cluster.argtypes = [vp, vp, i64, i32, vp, i32, vp, vp, C.POINTER(OfflineClusterConfig), i32, vp, vp, i32,
                                     C.POINTER(i32), C.POINTER(OfflineClusterInfo)]
    L.fa_offline_cluster_ex.argtypes = [vp, vp, i64, i32, vp, i32, vp, vp, C.POINTER(OfflineClusterConfig), i32, vp, vp, i32,
                                        C.POINTER(i32), C.POINTER(OfflineClusterInfo), vp, vp, vp]
    u64 = C.c_uint64
    L.fa_arpa_parse.argtypes = [vp, C.c_char_p, i64, C.POINTER(vp)]
    L.fa_arpa_destroy.argtypes = [vp]
    L.fa_arpa_destroy.restype = None
    L.fa_arpa_unigram_count.argtypes = [vp]
    L.fa_arpa_unigram_count.restype = i64
    L.fa_arpa_bigram_context_count.argtypes = [vp]
    L.fa_arpa_bigram_context_count.restype = i64
    L.fa_arpa_score.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(f32)]
    L.fa_ctc_vocab_create.argtypes = [vp, vp, vp, i32, i32, C.POINTER(vp)]
    L.fa_ctc_vocab_destroy.argtypes = [vp]
    L.fa_ctc_vocab_destroy.restype = None
    L.fa_ctc_beam_search_batch_dev.argtypes = [vp, vp, i32, i32, i32, i64, i64, vp, vp, vp, i32, f32, f32, i32, i32, vp, vp, vp]
    L.fa_ctc_beam_search_batch.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, f32, f32, i32, i32, vp, vp, vp]
    L.fa_wav_pcm16_size.argtypes = [i64]
    L.fa_wav_pcm16_size.restype = i64
    L.fa_wav_encode_pcm16.argtypes = [vp, vp, i64, f64, i32, vp, i64, C.POINTER(i64)]
    L.fa_wav_decode.argtypes = [vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    L.fa_rttm_parse.argtypes = [C.c_char_p, i64, i32, vp, i64, C.POINTER(i64), C.c_char_p, i64]
    L.fa_rttm_format.argtypes = [vp, i64, C.c_char_p, C.c_char_p, i64]
    L.fa_rttm_format.restype = i64
    L.fa_export_embeddings_json.argtypes = [vp, i64, vp, i32, vp, i32, vp, i64, C.c_char_p, i64]
    L.fa_export_embeddings_json.restype = i64
    L.fa_seeded_rng_next.argtypes = [C.POINTER(u64)]
    L.fa_seeded_rng_next.restype = u64
    L.fa_seeded_rng_below.argtypes = [C.POINTER(u64), u64]
    L.fa_seeded_rng_below.restype = u64
    L.fa_kmeans_cluster.argtypes = [vp, vp, i64, i32, i32, i32, u64, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.fa_kmeans_cluster_ninit.argtypes = [vp, vp, i64, i32, i32, i32, i32, u64, vp, vp, C.POINTER(i32), C.POINTER(i32), vp]
    L.fa_speaker_constraints_resolve.argtypes = [i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64 * 3)]
    L.fa_speaker_constraints_resolve.restype = None
    L.fa_resample_linear_frames.argtypes = [i64, f64, f64]
    L.fa_resample_linear_frames.restype = i64
    L.fa_resample_linear.argtypes = [vp, vp, i32, i64, f64, f64, vp, i64, C.POINTER(i64)]
    L.fa_resample_poly_frames.argtypes = [i64, i32, i32]
    L.fa_resample_poly_frames.restype = i64
    L.fa_resample_poly_taps.argtypes = [i32, i32, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.fa_resample_poly.argtypes = [vp, vp, i64, i32, i32, vp, i64, C.POINTER(i64)]
    L.fa_resample_poly_dev.argtypes = [vp, vp, i64, i32, i32, vp, i64, C.POINTER(i64)]
    L.fa_offline_cluster_batch.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, C.POINTER(OfflineClusterConfig), vp, vp, i32, vp, vp, vp]
    L.fa_offline_cluster_batch_dev.argtypes = L.fa_offline_cluster_batch.argtypes
    L.fa_device_count.argtypes = [C.POINTER(i32)]
    L.fa_pool_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.fa_pool_destroy.argtypes = [vp]
    L.fa_pool_destroy.restype = None
    L.fa_pool_size.argtypes = [vp]
    L.fa_pool_size.restype = i32
    L.fa_pool_context.argtypes = [vp, i32]
    L.fa_pool_context.restype = vp
    L.fa_ctx_device.argtypes = [vp]
    L.fa_ctx_device.restype = i32
    L.fa_pool_acquire.argtypes = [vp, C.POINTER(vp)]
    L.fa_pool_release.argtypes = [vp, vp]
    L.fa_pool_release.restype = None
    L.fa_mel_batch_sharded.argtypes = [vp, C.POINTER(MelConfig), vp, vp, i32, vp, vp, i32, vp, vp]
    L.fa_ctc_greedy_batch_sharded.argtypes = [vp, vp, i32, i32, i32, i32, i64, i64, vp, i32, vp, vp, vp]
    L.fa_ahc_linkage_many.argtypes = [vp, i32, vp, vp, sz, vp, i32, vp, vp]
    _lib = L
    return L


class Context:
    """fa_ctx: one device + one stream + cached workspaces (one per host thread)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        st = lib().fa_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h))
        if st != SUCCESS:
            raise FluidAudioHipError(st, "fa_ctx_create", "no usable MI355X visible" if st == RUNTIME_ERROR else "")
        self.device = device

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return lib().fa_ctx_stream(self._h) or 0

    def synchronize(self):
        self.check(lib().fa_ctx_synchronize(self._h), "fa_ctx_synchronize")

    @contextlib.contextmanager
    def torch_ordered(self, enabled: bool = True):
        """Stream ordering for the `*_dev` entries called with torch tensors: the context's stream first waits for
        everything already enqueued on torch's current stream (producers of the inputs, `.contiguous()` copies), and
        torch's current stream afterwards waits for what the body enqueued on the context's stream — so later torch ops
        (and the caching allocator's reuse of freed blocks) are ordered behind the kernels.  Two event record/wait pairs,
        no host synchronisation.  `enabled=False`: the caller orders the streams itself (bench.py's timed loop)."""
        if not enabled:
            yield
            return
        import torch
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            own = torch.cuda.ExternalStream(self.stream, device=self.device)
        if cur.cuda_stream == own.cuda_stream:
            yield
            return
        own.wait_stream(cur)
        try:
            yield
        finally:
            cur.wait_stream(own)

    def last_error(self) -> str:
        return (lib().fa_ctx_last_error(self._h) or b"").decode()

    # workspace policy (include/fluidaudio_hip.h): what the context may keep cached between calls / may take at all
    def set_workspace_limit(self, nbytes: int):
        self.check(lib().fa_ctx_set_workspace_limit(self._h, nbytes), "fa_ctx_set_workspace_limit")

    def set_workspace_cap(self, nbytes: int | None):
        self.check(lib().fa_ctx_set_workspace_cap(self._h, (1 << 64) - 1 if nbytes is None else nbytes), "fa_ctx_set_workspace_cap")

    def trim(self):
        self.check(lib().fa_ctx_trim(self._h), "fa_ctx_trim")

    def workspace_bytes(self) -> int:
        return int(lib().fa_ctx_workspace_bytes(self._h))

    def reserve(self, n_max: int, d: int, recordings: int = 1):
        """Take the linkage workspace of `recordings` problems of up to n_max x d now (server start-up), not inside the first request."""
        self.check(lib().fa_ctx_reserve(self._h, int(n_max), int(d), int(recordings)), "fa_ctx_reserve")

    def check(self, status: int, where: str):
        if status != SUCCESS:
            raise FluidAudioHipError(status, where, self.last_error())

    def sclk_mhz(self, spin_us: int = 200) -> float:
        """The shader clock right now (fa_debug_sclk_mhz): latency-bound legs print it next to their timing."""
        v = C.c_double()
        self.check(lib().fa_debug_sclk_mhz(self._h, spin_us, C.byref(v)), "fa_debug_sclk_mhz")
        return v.value

    def close(self):
        if self._h:
            lib().fa_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_array(shape, dtype):
    """numpy array over page-locked host memory from fa_host_alloc (freed when the array and its views are gone)."""
    import numpy as np
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    p = lib().fa_host_alloc(max(n * dt.itemsize, 1))
    if not p:
        raise MemoryError("fa_host_alloc failed")

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            lib().fa_host_free(self.ptr)

    buf = (C.c_char * max(n * dt.itemsize, 1)).from_address(p)
    buf._owner = _Owner(p)
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


_default_ctx: dict[int, Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("FLUIDAUDIO_HIP_USE_LOCAL_RANK") else 0
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
