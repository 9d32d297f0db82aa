"""Wire formats at the edges of the hot path over the C ABI (csrc/formats.hip): ``AudioWAV.data`` (reference:
Sources/FluidAudio/Shared/AudioConverter.swift:474-532), the RTTM loaders (Sources/FluidAudioCLI/Utils/RTTMParser.swift:22-63,
Sources/FluidAudioCLI/Commands/SortformerBenchmark.swift:681-731) and the embedding export
(Sources/FluidAudio/Diarizer/Offline/Core/OfflineDiarizerManager.swift:913-955)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


class RttmSegment(C.Structure):
    _fields_ = [("start_seconds", C.c_float), ("end_seconds", C.c_float), ("quality", C.c_float), ("speaker_id", C.c_char * 64)]


class ExportEmbedding(C.Structure):
    _fields_ = [("chunk_index", C.c_int32), ("speaker_index", C.c_int32), ("start_frame", C.c_int32), ("end_frame", C.c_int32),
                ("start_time", C.c_double), ("end_time", C.c_double)]


@dataclass
class TimedSpeakerSegment:
    speaker_id: str
    start_time_seconds: float
    end_time_seconds: float
    quality_score: float = 1.0


class RTTMParserError(ValueError):
    pass


class AudioWAV:
    @staticmethod
    def data(samples, sample_rate: float, normalize: bool = True, ctx: L.Context | None = None) -> bytes:
        """AudioWAV.data(from:sampleRate:normalize:) -> the bytes of a 16-bit PCM mono WAV file."""
        x = np.ascontiguousarray(samples, np.float32)
        ctx = ctx or L.default_context()
        out = np.zeros(L.lib().fa_wav_pcm16_size(x.size), np.uint8)
        n = C.c_int64()
        ctx.check(L.lib().fa_wav_encode_pcm16(ctx.handle, x.ctypes.data, x.size, float(sample_rate), int(bool(normalize)), out.ctypes.data,
                                              out.size, C.byref(n)), "fa_wav_encode_pcm16")
        return out[:n.value].tobytes()

    @staticmethod
    def read(data: bytes):
        """Extension: (samples float32 [frames, channels], sample_rate) of a 16-bit PCM or 32-bit float WAV."""
        buf = np.frombuffer(data, np.uint8)
        frames, ch, sr = C.c_int64(), C.c_int32(), C.c_int32()
        if L.lib().fa_wav_decode(buf.ctypes.data, buf.size, None, 0, C.byref(frames), C.byref(ch), C.byref(sr)) != 0:
            raise ValueError("not a PCM16 / float32 RIFF/WAVE file")
        out = np.zeros((frames.value, ch.value), np.float32)
        st = L.lib().fa_wav_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.size, None, None, None)
        assert st == 0
        return out, sr.value


class RTTMParser:
    @staticmethod
    def parse(text: str, strict: bool = True) -> list:
        raw = text.encode("utf-8")
        n = C.c_int64()
        bad = C.create_string_buffer(512)
        st = L.lib().fa_rttm_parse(raw, len(raw), int(strict), None, 0, C.byref(n), bad, 512)
        if st != 0:
            raise RTTMParserError("Invalid RTTM line: " + bad.value.decode("utf-8", "replace"))   # RTTMParser.swift:15
        segs = (RttmSegment * max(n.value, 1))()
        st = L.lib().fa_rttm_parse(raw, len(raw), int(strict), segs, n.value, C.byref(n), bad, 512)
        assert st == 0
        return [TimedSpeakerSegment(s.speaker_id.decode("utf-8"), float(s.start_seconds), float(s.end_seconds), float(s.quality))
                for s in segs[:n.value]]

    @staticmethod
    def load_segments(path: str) -> list:
        with open(path, encoding="utf-8") as fh:   # a missing file raises FileNotFoundError (RTTMParserError.fileNotFound)
            return RTTMParser.parse(fh.read(), strict=True)

    @staticmethod
    def format(segments, file_id: str = "audio") -> str:
        arr = (RttmSegment * max(len(segments), 1))()
        for a, s in zip(arr, segments):
            a.start_seconds, a.end_seconds, a.quality = s.start_time_seconds, s.end_time_seconds, s.quality_score
            a.speaker_id = s.speaker_id.encode("utf-8")[:63]
        n = L.lib().fa_rttm_format(arr, len(segments), file_id.encode(), None, 0)
        buf = C.create_string_buffer(n + 1)
        L.lib().fa_rttm_format(arr, len(segments), file_id.encode(), buf, n + 1)
        return buf.value.decode("utf-8")


def export_embeddings_json(items, embedding256, rho128, assignments) -> str:
    """exportEmbeddings (:913-955).  items: sequence of (chunkIndex, speakerIndex, startFrame, endFrame, startTime, endTime)."""
    n = len(items)
    arr = (ExportEmbedding * max(n, 1))()
    for a, it in zip(arr, items):
        a.chunk_index, a.speaker_index, a.start_frame, a.end_frame, a.start_time, a.end_time = it
    e = np.ascontiguousarray(embedding256, np.float32).reshape(n, -1) if n else np.zeros((0, 0), np.float32)
    r = np.ascontiguousarray(rho128, np.float64).reshape(n, -1) if n else np.zeros((0, 0))
    asg = np.ascontiguousarray(assignments, np.int32)
    args = (arr, n, e.ctypes.data, e.shape[1] if n else 0, r.ctypes.data, r.shape[1] if n else 0, asg.ctypes.data, asg.size)
    size = L.lib().fa_export_embeddings_json(*args, None, 0)
    buf = C.create_string_buffer(size + 1)
    L.lib().fa_export_embeddings_json(*args, buf, size + 1)
    return buf.value.decode("utf-8")
