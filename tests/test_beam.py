"""CTC prefix beam search + ARPA language model: the CPU restatement against the reference's own test answers
(CtcDecoderTests.swift:145-260, ARPALanguageModelTests.swift:39-176), the library's host-side LM against the restatement
(no GPU), and the device search against the restatement (gpu)."""
import numpy as np
import pytest

W = "\u2581"
SAMPLE_ARPA = ("\\data\\\nngram 1=4\nngram 2=2\n\n\\1-grams:\n-1.0\tthe\t-0.5\n-1.2\tcat\t-0.3\n-1.5\tsat\t0.0\n-2.0\t<unk>\t0.0\n\n"
               "\\2-grams:\n-0.5\tthe\tcat\n-0.8\tcat\tsat\n\n\\end\\\n")
LOG10 = float(np.float32(np.log(10.0)))


def test_oracle_beam_known_answers(oracle_mod):
    o = oracle_mod
    v = {0: W + "hello", 1: W + "world"}
    lp = [[0.0, -100.0, -100.0], [-100.0, -100.0, 0.0], [-100.0, 0.0, -100.0]]
    ids, _ = o.ctc_beam_search(lp, v, None, 5, 0.0, 0.0, 2)
    assert o.decode_ctc_token_ids(ids, v) == "hello world"                            # == greedy (:145-158)
    assert o.ctc_beam_search([[-100.0, 0.0]] * 3, {0: W + "hello"}, None, 5, 0.0, 0.0, 1)[0] == []      # all blanks (:160-172)
    assert o.ctc_beam_search([], {0: W + "hello"}, None, 5, 0.0, 0.0, 1)[0] == []                       # empty (:174-181)
    assert o.ctc_beam_search([[0.0, -100.0]], {0: W + "hello"}, None, 5, 0.0, 0.0, 1)[0] == [0]         # single token (:183-194)
    lm = o.ARPALanguageModel.parse(SAMPLE_ARPA)
    v = {0: W + "the", 1: W + "cat", 2: W + "dog"}
    lp = [[0.0, -100.0, -100.0, -100.0], [-100.0, -1.0, -0.9, -100.0]]
    assert o.decode_ctc_token_ids(o.ctc_beam_search(lp, v, None, 10, 0.0, 0.0, 3)[0], v) == "the dog"   # :148-165
    assert o.decode_ctc_token_ids(o.ctc_beam_search(lp, v, lm, 10, 5.0, 0.0, 3)[0], v) == "the cat"     # :167-175


def check_lm(lm, count_uni, count_ctx):
    assert count_uni == 4 and count_ctx == 2                                          # :41-47
    assert lm.score("cat", "the") == pytest.approx(-0.5 * LOG10, abs=1e-3)            # bigram (:101-108)
    assert lm.score("sat", "the") == pytest.approx(-0.5 * LOG10 - 1.5 * LOG10, abs=1e-3)   # backoff + unigram (:110-120)
    assert lm.score("cat", None) == pytest.approx(-1.2 * LOG10, abs=1e-3)             # :122-129
    assert lm.score("xyzzy", None) == pytest.approx(-23.026, abs=1e-3)                # :131-137
    assert lm.score("xyzzy", "the") == pytest.approx(-0.5 * LOG10 - 23.026, abs=1e-3)   # :139-148


def test_arpa_oracle_and_library_agree(fa, oracle_mod):
    ref = oracle_mod.ARPALanguageModel.parse(SAMPLE_ARPA)
    check_lm(ref, len(ref.unigrams), len(ref.bigrams))
    lm = fa.ARPALanguageModel(SAMPLE_ARPA)                                            # parsing and scoring are host code
    check_lm(lm, lm.unigram_count, lm.bigram_context_count)
    words = ["the", "cat", "sat", "<unk>", "dog", ""]
    for w in words:
        for p in words + [None]:
            assert lm.score(w, p) == float(ref.score(w, p)), (w, p)                   # bit-identical float arithmetic
    empty = fa.ARPALanguageModel("\\data\\\nngram 1=0\n\n\\1-grams:\n\n\\end\\\n")    # :83-97
    assert empty.unigram_count == 0 and empty.bigram_context_count == 0
    with pytest.raises(fa.ARPAError):                                                 # :77-80
        fa.ARPALanguageModel.load("/nonexistent/path/to/model.arpa")
    # malformed lines are skipped, later sections ignored, duplicate entries overwrite (:52-85)
    odd = ("\\data\\\nngram 1=2\n\\1-grams:\nnot_a_number\tw\n-1.0\ta\n-2.0\ta\t-0.25\n-1.0 \tb\n\\2-grams:\n-0.1\ta\tb\textra\n-0.3\ta\n"
           "\\3-grams:\n-0.1\ta\tb\tc\n\\end\\\n-9.0\tz\n")
    r2, l2 = oracle_mod.ARPALanguageModel.parse(odd), fa.ARPALanguageModel(odd)
    assert l2.unigram_count == len(r2.unigrams) and l2.bigram_context_count == len(r2.bigrams)
    for w in ["a", "b", "z", "c"]:
        for p in ["a", "b", None]:
            assert l2.score(w, p) == float(r2.score(w, p)), (w, p)


# ---- CtcDecoderDemoTests.swift: the reference's three demo cases (greedy vs beam vs beam + LM) ---------------------------------------
DEMO1_VOCAB = {0: W + "patient", 1: W + "has", 2: W + "diabetes", 3: W + "die", 4: W + "beetus", 5: W + "high", 6: W + "blood", 7: W + "pressure"}


def _demo_rows(n, *hot):
    """frames of CtcDecoderDemoTests.swift:27-53 / :99-113: -10 everywhere except the listed (index, value) pairs"""
    r = [-10.0] * n
    for i, v in hot:
        r[i] = v
    return r


DEMO1_LP = ([_demo_rows(9, (0, -1.0))] * 2 + [_demo_rows(9, (8, -1.0))] + [_demo_rows(9, (1, -1.0))] * 2 + [_demo_rows(9, (8, -1.0))] +
            [_demo_rows(9, (2, -1.5), (3, -1.4))] * 2 + [_demo_rows(9, (8, -1.0))] + [_demo_rows(9, (4, -1.2))] * 2)
DEMO2_VOCAB = {0: W + "the", 1: W + "cat", 2: W + "sat", 3: W + "dog"}
DEMO2_LP = ([_demo_rows(5, (0, -1.0))] * 2 + [_demo_rows(5, (4, -1.0))] + [_demo_rows(5, (1, -1.5), (3, -1.4))] * 2 + [_demo_rows(5, (4, -1.0))] +
            [_demo_rows(5, (2, -1.0))] * 2)
# the demo models are built in code from NATURAL-log entries (ARPALanguageModel.Entry(logProb:backoff:), :80-87, :163-196)
MEDICAL_UNI = {"patient": (-1.5, -0.3), "has": (-1.8, -0.2), "diabetes": (-2.2, -0.1), "hypertension": (-2.5, -0.1), "die": (-4.0, -0.5),
               "beetus": (-5.0, -0.5), "hyper": (-4.5, -0.4), "tension": (-4.3, -0.4)}
MEDICAL_BI = {"patient": {"has": (-0.3, 0.0)}, "has": {"diabetes": (-0.5, 0.0), "hypertension": (-0.6, 0.0)}}
CATDOG_UNI = {"the": (-1.0, -0.3), "cat": (-1.5, 0.0), "dog": (-1.5, 0.0), "sat": (-1.5, 0.0)}
CATDOG_BI = {"the": {"cat": (-0.3, 0.0), "dog": (-2.0, 0.0)}}


def oracle_lm(oracle_mod, uni, bi):
    lm = oracle_mod.ARPALanguageModel()
    lm.unigrams = {w: (np.float32(p), np.float32(b)) for w, (p, b) in uni.items()}
    lm.bigrams = {c: {w: (np.float32(p), np.float32(b)) for w, (p, b) in d.items()} for c, d in bi.items()}
    return lm


def _log10_text(v):
    """a decimal log10 value whose float32 parse times float32(ln 10) rounds to float32(v) — the library takes models as ARPA text
    (log10), the demo tests hand over natural-log entries"""
    want, k = np.float32(v), np.float32(np.log(10.0))
    c = np.float32(want / k)
    for cand in (c, np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf))):
        if np.float32(cand * k) == want:
            return repr(float(cand))
    return None


def arpa_text(uni, bi):
    """ARPA text of a model given in natural-log entries; (text, exact): exact = every entry reproduces its float32 value bit for bit"""
    exact, lines = True, ["\\data\\", "", "\\1-grams:"]
    def field(v):
        nonlocal exact
        t = _log10_text(v)
        if t is None:
            exact, t = False, repr(float(np.float32(v) / np.float32(np.log(10.0))))
        return t
    for w, (p, b) in uni.items():
        lines.append(f"{field(p)}\t{w}\t{field(b)}")
    lines += ["", "\\2-grams:"]
    for c, d in bi.items():
        for w, (p, b) in d.items():
            lines.append(f"{field(p)}\t{c}\t{w}")
    lines += ["", "\\end\\", ""]
    return "\n".join(lines), exact


def test_oracle_reference_demo_cases(oracle_mod):
    """CtcDecoderDemoTests.swift:11-76 and :80-135: greedy and beam search without a model follow the acoustics ("die beetus", "dog"),
    the word-level model turns the search to the real words."""
    o = oracle_mod
    ids = o.ctc_greedy(np.asarray(DEMO1_LP, np.float32), 8)
    assert o.decode_ctc_token_ids(ids, DEMO1_VOCAB) == "patient has die beetus"                                     # :56-58
    assert o.decode_ctc_token_ids(o.ctc_beam_search(DEMO1_LP, DEMO1_VOCAB, None, 10, 0.3, 0.0, 8)[0], DEMO1_VOCAB) == "patient has die beetus"   # :61-66
    lm = oracle_lm(o, MEDICAL_UNI, MEDICAL_BI)
    assert o.decode_ctc_token_ids(o.ctc_beam_search(DEMO1_LP, DEMO1_VOCAB, lm, 10, 5.0, 0.0, 8)[0], DEMO1_VOCAB) == "patient has diabetes"       # :69-76
    assert o.decode_ctc_token_ids(o.ctc_greedy(np.asarray(DEMO2_LP, np.float32), 4), DEMO2_VOCAB) == "the dog sat"    # :115-118
    lm2 = oracle_lm(o, CATDOG_UNI, CATDOG_BI)
    assert o.decode_ctc_token_ids(o.ctc_beam_search(DEMO2_LP, DEMO2_VOCAB, lm2, 10, 2.0, 0.0, 4)[0], DEMO2_VOCAB) == "the cat sat"               # :131-137


def test_demo_models_as_arpa_text_and_windows_line_endings(fa, oracle_mod):
    """The library takes a model as ARPA text: the demo models written as log10 text reproduce the natural-log entries (host code, no
    GPU), and a file with \\r\\n line endings parses like the reference's reader, which trims every line (CtcDecoderDemoTests.swift:139-159)."""
    for uni, bi in ((MEDICAL_UNI, MEDICAL_BI), (CATDOG_UNI, CATDOG_BI)):
        text, exact = arpa_text(uni, bi)
        ref, lib, parsed = oracle_lm(oracle_mod, uni, bi), fa.ARPALanguageModel(text), oracle_mod.ARPALanguageModel.parse(text)
        assert lib.unigram_count == len(uni) and lib.bigram_context_count == len(bi)
        words = list(uni) + ["xyzzy"]
        for w in words:
            for p in words + [None]:
                assert lib.score(w, p) == float(parsed.score(w, p))
                if exact:
                    assert lib.score(w, p) == float(ref.score(w, p)), (w, p)
                else:
                    assert lib.score(w, p) == pytest.approx(float(ref.score(w, p)), rel=3e-7)
    crlf = "\\data\\\r\nngram 1=2\r\n\r\n\\1-grams:\r\n-1.0\thello\t0.0\r\n-1.0\tworld\t0.0\r\n\r\n\\end\\\r\n"
    ref, lib = oracle_mod.ARPALanguageModel.parse(crlf), fa.ARPALanguageModel(crlf)
    assert len(ref.unigrams) == 2 and lib.unigram_count == 2 and {"hello", "world"} == set(ref.unigrams)
    assert lib.score("hello", None) == float(ref.score("hello", None)) == pytest.approx(-LOG10, rel=1e-6)
    assert lib.score("world", "hello") == float(ref.score("world", "hello"))


def test_reference_arpa_fixture_parses_identically(fa, oracle_mod):
    """Tests/.../CTC/sample_medical.arpa, the reference's ARPA fixture (read where it lies; not copied): 15 unigrams, 12 bigrams in 5
    contexts; library and restatement give the same float for every (word, context) pair."""
    import os
    path = "/root/reference/Tests/FluidAudioTests/ASR/Parakeet/SlidingWindow/CTC/sample_medical.arpa"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    text = open(path, encoding="utf-8").read()
    ref, lib = oracle_mod.ARPALanguageModel.parse(text), fa.ARPALanguageModel.load(path)
    assert len(ref.unigrams) == lib.unigram_count == 15
    assert len(ref.bigrams) == lib.bigram_context_count == 5 and sum(len(d) for d in ref.bigrams.values()) == 12
    words = list(ref.unigrams) + ["beetus"]
    for w in words:
        for p in words + [None]:
            assert lib.score(w, p) == float(ref.score(w, p)), (w, p)
    assert lib.score("pressure", "blood") == pytest.approx(-0.1 * LOG10, rel=1e-6)
    assert lib.score("diabetes", "doctor") == pytest.approx((-0.2 - 2.2) * LOG10, rel=1e-6)      # backoff of "doctor" + unigram


def random_case(rng, T, V, peaky):
    x = rng.standard_normal((T, V)).astype(np.float32) * peaky
    x = x - np.log(np.exp(x.astype(np.float64)).sum(1, keepdims=True)).astype(np.float32)
    return x.astype(np.float32)


def make_vocab(rng, V, blank):
    words = ["the", "cat", "sat", "dog", "on", "mat", "a"]
    voc = {}
    for v in range(V):
        if v == blank:
            continue
        kind = rng.integers(0, 4)
        if kind == 0:
            voc[v] = W + words[rng.integers(0, len(words))]
        elif kind == 1:
            voc[v] = W + "c"
        elif kind == 2:
            voc[v] = ["at", "s", "og", "t", "he"][rng.integers(0, 5)]
        # kind 3: id missing from the vocabulary -> "" piece
    voc[(blank + 1) % V] = W                                                          # bare boundary piece: empty word start
    return voc


@pytest.mark.gpu
def test_reference_cases_on_device(fa, gpu_ctx):
    v = {0: W + "hello", 1: W + "world"}
    lp = [[0.0, -100.0, -100.0], [-100.0, -100.0, 0.0], [-100.0, 0.0, -100.0]]
    assert fa.ctc_beam_search(lp, v, None, 5, 0.0, 0.0, 2, ctx=gpu_ctx) == fa.ctc_greedy_decode(lp, v, blank_id=2) == "hello world"
    assert fa.ctc_beam_search([[-100.0, 0.0]] * 3, {0: W + "hello"}, None, 5, 0.0, 0.0, 1, ctx=gpu_ctx) == ""
    assert fa.ctc_beam_search([], {0: W + "hello"}, None, 5, 0.0, 0.0, 1, ctx=gpu_ctx) == ""
    assert fa.ctc_beam_search([[0.0, -100.0]], {0: W + "hello"}, None, 5, 0.0, 0.0, 1, ctx=gpu_ctx) == "hello"
    lm = fa.ARPALanguageModel(SAMPLE_ARPA, ctx=gpu_ctx)
    v = {0: W + "the", 1: W + "cat", 2: W + "dog"}
    lp = [[0.0, -100.0, -100.0, -100.0], [-100.0, -1.0, -0.9, -100.0]]
    assert fa.ctc_beam_search(lp, v, None, 10, 0.0, 0.0, 3, ctx=gpu_ctx) == "the dog"
    assert fa.ctc_beam_search(lp, v, lm, 10, 5.0, 0.0, 3, ctx=gpu_ctx) == "the cat"


@pytest.mark.gpu
def test_reference_demo_cases_on_device(fa, gpu_ctx, oracle_mod):
    """CtcDecoderDemoTests.swift:11-76, :80-135 on the device: the reference's expected strings, and token ids + scores of the restatement."""
    assert fa.ctc_greedy_decode(DEMO1_LP, DEMO1_VOCAB, blank_id=8, ctx=gpu_ctx) == "patient has die beetus"
    assert fa.ctc_beam_search(DEMO1_LP, DEMO1_VOCAB, None, 10, 0.3, 0.0, 8, ctx=gpu_ctx) == "patient has die beetus"
    text, exact = arpa_text(MEDICAL_UNI, MEDICAL_BI)
    assert exact
    lm = fa.ARPALanguageModel(text, ctx=gpu_ctx)
    assert fa.ctc_beam_search(DEMO1_LP, DEMO1_VOCAB, lm, 10, 5.0, 0.0, 8, ctx=gpu_ctx) == "patient has diabetes"
    assert fa.ctc_greedy_decode(DEMO2_LP, DEMO2_VOCAB, blank_id=4, ctx=gpu_ctx) == "the dog sat"
    text2, exact2 = arpa_text(CATDOG_UNI, CATDOG_BI)
    assert exact2
    lm2 = fa.ARPALanguageModel(text2, ctx=gpu_ctx)
    assert fa.ctc_beam_search(DEMO2_LP, DEMO2_VOCAB, lm2, 10, 2.0, 0.0, 4, ctx=gpu_ctx) == "the cat sat"
    for lp, voc, dev_lm, uni, bi, w, blank in ((DEMO1_LP, DEMO1_VOCAB, lm, MEDICAL_UNI, MEDICAL_BI, 5.0, 8), (DEMO2_LP, DEMO2_VOCAB, lm2, CATDOG_UNI, CATDOG_BI, 2.0, 4)):
        x = np.asarray(lp, np.float32)
        ids, scores = fa.ctc_beam_search_ids_batch(x[None], voc, dev_lm, 10, w, 0.0, blank, ctx=gpu_ctx)
        want, total = oracle_mod.ctc_beam_search(x, voc, oracle_lm(oracle_mod, uni, bi), 10, w, 0.0, blank)
        assert ids[0] == want and scores[0] == pytest.approx(total, rel=2e-6, abs=2e-5)


@pytest.mark.gpu
def test_batches_beyond_one_trie_allocation_are_walked_in_pieces(fa, gpu_ctx):
    """The prefix tries of a launch are capped at 2 GiB: 2 049 frames x beam 128 need 8 MB per utterance, so 258 utterances go through TWO launches
    (256 + 2), each with its own top-token pre-pass and table.  Utterances on both sides of the cut must decode exactly as they do alone."""
    rng = np.random.default_rng(77)
    B, T, V, blank = 258, 2049, 5, 4
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    x = (x - np.log(np.exp(x.astype(np.float64)).sum(2, keepdims=True))).astype(np.float32)
    valid = [T] * B
    valid[255] = T - 7
    valid[257] = 11
    ids, scores = fa.ctc_beam_search_ids_batch(x, None, None, 128, 0.0, 0.0, blank, 3, valid_frames=valid, ctx=gpu_ctx)
    for b in (0, 255, 256, 257):
        one, sc = fa.ctc_beam_search_ids_batch(x[b:b + 1], None, None, 128, 0.0, 0.0, blank, 3, valid_frames=valid[b:b + 1], ctx=gpu_ctx)
        assert ids[b] == one[0] and (len(ids[b]) > 0 or b == 257), b
        assert scores[b] == sc[0]


@pytest.mark.gpu
@pytest.mark.parametrize("T,V,K,levels,seed", [(30, 200, 40, 6, 0), (20, 1025, 40, 3, 1), (16, 1025, 64, 40, 2), (10, 3000, 33, 4, 3), (25, 70, 64, 5, 4),
                                               (12, 1088, 17, 2, 5), (8, 1089, 40, 2, 6), (9, 12, 0, 3, 7), (9, 64, 64, 2, 8), (6, 65, 64, 2, 9)])
def test_top_token_ties_follow_the_index(fa, gpu_ctx, oracle_mod, T, V, K, levels, seed):
    """The frame's token candidates are `sorted { frame[$0] > frame[$1] }.prefix(tokenCandidates)` (CtcDecoder.swift:141-144, stable: equal
    log-probabilities keep index order).  Log-probabilities quantised to a few levels — with runs of -inf — put hundreds of exact ties across
    the K-th place of every frame; the pre-pass (ctc_topk_kernel, one wavefront per frame: registers up to 1 088 tokens, re-read rows above)
    must offer the walk the same candidates in the same order as the restatement."""
    rng = np.random.default_rng(100 + seed)
    blank = V - 1 if seed % 2 == 0 else 3
    x = -rng.integers(1, levels + 1, size=(2, T, V)).astype(np.float32) * 0.75
    x[rng.random((2, T, V)) < 0.1] = -np.inf
    x[:, :, blank] = -1.0
    x[0, T // 2, :] = -2.0                                              # one frame with every token tied
    ids, scores = fa.ctc_beam_search_ids_batch(x, None, None, 12, 0.0, 0.0, blank, K, ctx=gpu_ctx)
    for b in range(2):
        want, total = oracle_mod.ctc_beam_search(x[b], {}, None, 12, 0.0, 0.0, blank, K)
        assert ids[b] == want, (b, ids[b], want)


@pytest.mark.gpu
@pytest.mark.parametrize("T,V,W_,K,peaky,use_lm,seed", [(40, 12, 4, 3, 2.0, False, 0), (60, 33, 8, 6, 3.0, True, 1), (25, 9, 16, 8, 1.0, True, 2),
                                                       (120, 70, 10, 40, 4.0, True, 3), (80, 40, 100, 40, 2.5, False, 4), (30, 300, 5, 64, 3.0, True, 5),
                                                       (50, 17, 128, 16, 0.5, True, 6), (90, 6, 3, 5, 0.7, True, 7), (200, 8, 6, 4, 1.0, True, 8),
                                                       (150, 5, 12, 4, 0.3, False, 9), (64, 1025, 100, 40, 5.0, True, 10),
                                                       (20, 2000, 60, 30, 3.0, True, 12),     # 1 280 < V <= 4 160: keys staged in LDS, 17 per thread
                                                       (12, 5000, 40, 20, 3.0, False, 13)])   # V > 4 160: keys read from HBM, generic selection
def test_device_matches_restatement(fa, gpu_ctx, oracle_mod, T, V, W_, K, peaky, use_lm, seed):
    rng = np.random.default_rng(seed)
    blank = V - 1 if seed % 2 == 0 else 0
    B = 3
    batch = np.stack([random_case(rng, T, V, peaky) for _ in range(B)])
    voc = make_vocab(rng, V, blank)
    lm_ref = oracle_mod.ARPALanguageModel.parse(SAMPLE_ARPA) if use_lm else None
    lm_dev = fa.ARPALanguageModel(SAMPLE_ARPA, ctx=gpu_ctx) if use_lm else None
    valid = [T, T // 2, 0]
    ids, scores = fa.ctc_beam_search_ids_batch(batch, voc, lm_dev, W_, 0.7, 0.25, blank, K, valid_frames=valid, ctx=gpu_ctx)
    for b in range(B):
        want, total = oracle_mod.ctc_beam_search(batch[b, :valid[b]], voc, lm_ref, W_, 0.7, 0.25, blank, K)
        assert ids[b] == want, (b, ids[b], want)
        if total is not None:
            assert scores[b] == pytest.approx(total, rel=2e-6, abs=2e-5)


@pytest.mark.gpu
def test_limits_and_argument_errors(fa, gpu_ctx):
    """Largest supported beam / candidate counts on a TDT-sized vocabulary: structural sanity (no oracle at this size), the
    greedy path is among the hypotheses so the beam score can only be better, and out-of-range parameters are refused."""
    rng = np.random.default_rng(3)
    T, V, blank = 120, 8193, 8192
    x = random_case(rng, T, V, 4.0)
    ids, scores = fa.ctc_beam_search_ids_batch(x[None], None, None, 128, 0.0, 0.0, blank, 64, ctx=gpu_ctx)
    assert len(ids[0]) <= T and all(0 <= t < V and t != blank for t in ids[0]) and np.isfinite(scores[0])
    greedy_path = float(x.max(axis=1).sum())                      # log-prob of the best single alignment
    assert scores[0] >= greedy_path - 1e-3                        # the prefix total sums over alignments incl. that one
    ids1, _ = fa.ctc_beam_search_ids_batch(x[None], None, None, 1, 0.0, 0.0, blank, 1, ctx=gpu_ctx)   # beam 1, one candidate
    assert len(ids1[0]) <= T
    for bw, k in ((0, 40), (129, 40), (100, 65), (100, -1)):
        with pytest.raises(fa.FluidAudioHipError):
            fa.ctc_beam_search_ids_batch(x[None], None, None, bw, 0.0, 0.0, blank, k, ctx=gpu_ctx)
