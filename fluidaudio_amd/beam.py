"""Host-side mirror of ``ctcBeamSearch`` and ``ARPALanguageModel`` (reference:
Sources/FluidAudio/ASR/Parakeet/SlidingWindow/CTC/CtcDecoder.swift:72-241, .../CTC/ARPALanguageModel.swift:16-147) over the
HIP C ABI (csrc/beam.hip): one workgroup per utterance, the language model's hash tables resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .ctc import decode_ctc_token_ids


class ARPAError(OSError):
    """ARPAError.cannotOpen"""


class ARPALanguageModel:
    log10_to_nat = float(np.float32(np.log(10.0)))
    unk_log_prob = float(np.float32(-23.026))

    def __init__(self, text: str = "", ctx: L.Context | None = None):
        raw = text.encode("utf-8")            # parsing and score() are host code: no device context needed until a search
        h = C.c_void_p()
        st = L.lib().fa_arpa_parse(ctx.handle if ctx else None, raw, len(raw), C.byref(h))
        if st != 0:
            raise L.FluidAudioHipError(st, "fa_arpa_parse")
        self.handle = h

    @classmethod
    def load(cls, path: str, ctx: L.Context | None = None) -> "ARPALanguageModel":
        try:
            with open(path, "rb") as fh:
                data = fh.read()
        except OSError as e:
            raise ARPAError(f"Cannot open ARPA file: {path}") from e                 # ARPALanguageModel.swift:110-118
        return cls(data.decode("utf-8", "replace"), ctx)

    @property
    def unigram_count(self) -> int:
        return int(L.lib().fa_arpa_unigram_count(self.handle))

    @property
    def bigram_context_count(self) -> int:
        return int(L.lib().fa_arpa_bigram_context_count(self.handle))

    def score(self, word: str, prev: str | None = None) -> float:
        out = C.c_float()
        st = L.lib().fa_arpa_score(self.handle, word.encode("utf-8"), None if prev is None else prev.encode("utf-8"), C.byref(out))
        assert st == 0
        return float(out.value)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            L.lib().fa_arpa_destroy(h)


class CtcVocabulary:
    """[Int: String] token table uploaded for word tracking."""

    def __init__(self, vocabulary: dict, vocab_size: int, ctx: L.Context | None = None):
        self._ctx = ctx or L.default_context()
        items = sorted((int(k), v) for k, v in vocabulary.items())
        ids = (C.c_int32 * max(len(items), 1))(*[k for k, _ in items])
        pieces = (C.c_char_p * max(len(items), 1))(*[v.encode("utf-8") for _, v in items])
        h = C.c_void_p()
        self._ctx.check(L.lib().fa_ctc_vocab_create(self._ctx.handle, ids, pieces, len(items), int(vocab_size), C.byref(h)), "fa_ctc_vocab_create")
        self.handle, self.vocab_size = h, int(vocab_size)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            L.lib().fa_ctc_vocab_destroy(h)


def ctc_beam_search_ids_batch(log_probs, vocabulary: dict | None = None, lm: ARPALanguageModel | None = None, beam_width: int = 100,
                              lm_weight: float = 0.3, word_bonus: float = 0.0, blank_id: int = 1024, token_candidates: int = 40,
                              valid_frames=None, ctx: L.Context | None = None):
    """Batch of [B, T, V] float32 log-probabilities -> (list of token-id lists, scores float32[B])."""
    x = np.ascontiguousarray(log_probs, np.float32)
    assert x.ndim == 3
    B, T, V = x.shape
    if B == 0 or V == 0:
        return [[] for _ in range(B)], np.zeros(B, np.float32)
    ctx = ctx or L.default_context()
    voc = CtcVocabulary(vocabulary or {}, V, ctx) if lm is not None else None
    tokens = np.zeros((B, max(T, 1)), np.int32)
    lens = np.zeros(B, np.int32)
    scores = np.zeros(B, np.float32)
    vf = None if valid_frames is None else np.ascontiguousarray(valid_frames, np.int32)
    ctx.check(L.lib().fa_ctc_beam_search_batch(ctx.handle, x.ctypes.data, B, T, V, None if vf is None else vf.ctypes.data,
                                               voc.handle if voc else None, lm.handle if lm is not None else None, beam_width, lm_weight,
                                               word_bonus, blank_id, token_candidates, tokens.ctypes.data, lens.ctypes.data,
                                               scores.ctypes.data), "fa_ctc_beam_search_batch")
    return [tokens[b, :lens[b]].tolist() for b in range(B)], scores


def ctc_beam_search(log_probs, vocabulary: dict, lm: ARPALanguageModel | None = None, beam_width: int = 100, lm_weight: float = 0.3,
                    word_bonus: float = 0.0, blank_id: int = 1024, token_candidates: int = 40, ctx: L.Context | None = None) -> str:
    """ctcBeamSearch(logProbs:vocabulary:lm:beamWidth:lmWeight:wordBonus:blankId:tokenCandidates:) -> decoded text."""
    x = np.asarray(log_probs, np.float32)
    if x.size == 0:
        return ""                                                                    # guards (:129-131)
    ids, _ = ctc_beam_search_ids_batch(x[None], vocabulary, lm, beam_width, lm_weight, word_bonus, blank_id, token_candidates, ctx=ctx)
    return decode_ctc_token_ids(ids[0], vocabulary)
