"""fluidaudio_amd — MI355X-native replacement for FluidAudio's host-side hot path
(STFT->mel featurizer, argmax + CTC greedy collapse, centroid-linkage AHC, VBx) behind the
reference's own interfaces.  All arithmetic runs in csrc/libfluidaudio_hip.so (hand-written HIP
for gfx950) through the C ABI of include/fluidaudio_hip.h; there is no CPU fallback.
"""
from ._lib import (AHC_MODE_AUTO, AHC_MODE_EXACT, AHC_MODE_REFERENCE_ORDER, ALLOCATION_FAILURE, INVALID_ARGUMENT, RUNTIME_ERROR, SUCCESS, Context, FluidAudioHipError, build, default_context, lib)  # noqa: F401
from .ahc import AHCClustering, check_dendrogram, cut, fastcluster_compute_centroid_linkage, linkage, linkage_batch  # noqa: F401
from .beam import ARPAError, ARPALanguageModel, CtcVocabulary, ctc_beam_search, ctc_beam_search_ids_batch  # noqa: F401
from .ctc import (LogitsArgmax, ctc_greedy_decode, ctc_greedy_ids_batch, ctc_greedy_rows, ctc_greedy_ids_dev, ctc_log_probs_dev,  # noqa: F401
                  decode_ctc_token_ids)
from .formats import AudioWAV, RTTMParser, RTTMParserError, TimedSpeakerSegment, export_embeddings_json  # noqa: F401
from .kmeans import KMeansClustering, SeededRNG, SpeakerCountConstraints  # noqa: F401
from .mel import AudioMelSpectrogram, LuxTtsMelExtractor, MelPlan, UnifiedMelExtractor  # noqa: F401
from .pipeline import (ClusteringResult, OfflineClusteringConfig, cluster_embeddings, cluster_embeddings_batch, cluster_embeddings_stagewise,  # noqa: F401
                       select_training_embeddings)
from .pool import Pool, device_count  # noqa: F401
from .post import (ConstrainedClusterAssignment, HungarianAssignment, assign_embeddings, centroid_scores,  # noqa: F401
                   compute_centroids)
from .resample import linear_resample, poly_taps, resample_poly  # noqa: F401
from .sharding import gather_ragged_int32, shard_offsets, shard_range  # noqa: F401
from .tdt import (TdtConfig, TdtDurationMapping, TdtFrameNavigation, decode_logits as tdt_decode_logits,  # noqa: F401
                  decode_tables as tdt_decode_tables)
from .vbx import VBxClustering, VBxOutput  # noqa: F401

__version__ = "0.1.0"
