"""Wire formats: the host-only functions of the library (RTTM, embedding JSON, WAV reader) against Python restatements of the
reference (no GPU), and the device WAV writer against the float32 restatement (gpu)."""
import json
import struct

import numpy as np
import pytest

RTTM = """# comment line
SPEAKER meeting1 1 12.50 3.25 <NA> <NA> spk_B <NA> <NA>
   SPEAKER meeting1 1 0.000   1.5 <NA> <NA> spk_A <NA> <NA>

SPEAKER\tmeeting1\t1\t4.75\t0.333\t<NA>\t<NA>\tspk_A\t<NA>\t<NA>
SPEAKER meeting1 1 4.75 2.0 <NA> <NA> spk_C
"""


def test_rttm_strict_matches_restatement(fa, oracle_mod):
    got = fa.RTTMParser.parse(RTTM, strict=True)
    want = oracle_mod.rttm_parse(RTTM, strict=True)
    assert [(s.speaker_id, s.start_time_seconds, s.end_time_seconds) for s in got] == want
    assert [s.speaker_id for s in got] == ["spk_A", "spk_A", "spk_C", "spk_B"]       # sorted by start, stable for ties (:62)
    assert all(s.quality_score == 1.0 for s in got)
    assert got[1].end_time_seconds == float(np.float32(np.float32(4.75) + np.float32(0.333)))   # Float arithmetic (:49)


@pytest.mark.parametrize("bad", ["SPEAKER a 1 x 1.0 <NA> <NA> s", "LEXEME a 1 0.0 1.0 <NA> <NA> s <NA>", "SPEAKER a 1 0.0 1.0 <NA> <NA>",
                                 "SPEAKER a 1 0.0 1.0abc <NA> <NA> s"])
def test_rttm_invalid_lines(fa, oracle_mod, bad):
    text = "SPEAKER a 1 1.0 1.0 <NA> <NA> s1\n" + bad + "\nSPEAKER a 1 0.5 1.0 <NA> <NA> s2\n"
    with pytest.raises(fa.RTTMParserError) as e:                                       # RTTMParser.swift:37-46
        fa.RTTMParser.parse(text, strict=True)
    assert bad in str(e.value)
    with pytest.raises(ValueError):
        oracle_mod.rttm_parse(text, strict=True)
    lenient = fa.RTTMParser.parse("# c\n" + text, strict=False)                       # SortformerBenchmark.swift:700-714: skipped
    assert [(s.speaker_id, s.start_time_seconds, s.end_time_seconds) for s in lenient] == oracle_mod.rttm_parse("# c\n" + text, strict=False)
    assert [s.speaker_id for s in lenient] == ["s1", "s2"]                             # file order, not sorted


def test_rttm_empty_and_roundtrip(fa):
    assert fa.RTTMParser.parse("", strict=True) == [] and fa.RTTMParser.parse("\n\n# x\n", strict=True) == []
    segs = [fa.TimedSpeakerSegment("A", 0.0, 1.5), fa.TimedSpeakerSegment("B", 1.25, 4.0)]
    text = fa.RTTMParser.format(segs, "rec1")
    assert text.splitlines()[0] == "SPEAKER rec1 1 0.000 1.500 <NA> <NA> A <NA> <NA>"
    assert [(s.speaker_id, s.start_time_seconds, s.end_time_seconds) for s in fa.RTTMParser.parse(text)] == [("A", 0.0, 1.5), ("B", 1.25, 4.0)]


def test_rttm_format_long_names(fa):
    """Extension entry: a file id longer than any line buffer and a speaker id that fills its 64-byte field neither truncate the line
    nor read past the field."""
    segs = [fa.TimedSpeakerSegment("A" * 80, 0.0, 1.5), fa.TimedSpeakerSegment("B", 1.25, 4.0)]
    text = fa.RTTMParser.format(segs, "rec" * 200)
    lines = text.splitlines()
    assert len(lines) == 2 and all(ln.startswith("SPEAKER " + "rec" * 200 + " 1 ") and ln.endswith(" <NA> <NA>") for ln in lines)
    back = fa.RTTMParser.parse(text)
    assert [(s.speaker_id, s.start_time_seconds, s.end_time_seconds) for s in back] == [("A" * 63, 0.0, 1.5), ("B", 1.25, 4.0)]


def test_export_embeddings_json(fa):
    rng = np.random.default_rng(0)
    items = [(0, 1, 0, 99, 0.0, 1.98), (3, 0, 300, 431, 6.0, 8.625), (7, 2, 700, 800, 14.0, 16.1)]
    e = rng.standard_normal((3, 256)).astype(np.float32)
    r = rng.standard_normal((3, 128))
    text = fa.export_embeddings_json(items, e, r, [2, 0])                              # third entry has no assignment -> -1 (:932-934)
    doc = json.loads(text)
    assert [d["cluster"] for d in doc] == [2, 0, -1]
    keys = ["chunkIndex", "speakerIndex", "startFrame", "endFrame", "startTime", "endTime", "embedding256", "rho128", "cluster"]
    assert all(sorted(d) == sorted(keys) for d in doc)
    for d, it, ev, rv in zip(doc, items, e, r):
        assert (d["chunkIndex"], d["speakerIndex"], d["startFrame"], d["endFrame"], d["startTime"], d["endTime"]) == it
        assert np.array_equal(np.asarray(d["embedding256"], np.float32), ev)           # shortest round-trip text
        assert np.array_equal(np.asarray(d["rho128"], np.float64), rv)
    assert fa.export_embeddings_json([], np.zeros((0, 256)), np.zeros((0, 128)), []) == "[]"


def test_wav_reader_pcm16_and_float(fa, oracle_mod):
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 0.25], np.float32)
    data = oracle_mod.wav_pcm16(x, 16000, normalize=False)
    y, sr = fa.AudioWAV.read(data)
    assert sr == 16000 and y.shape == (6, 1)
    np.testing.assert_array_equal(y[:, 0], np.trunc(x * np.float32(32767)).astype(np.int16).astype(np.float32) / 32768.0)
    f = np.arange(8, dtype=np.float32).reshape(4, 2) / 8
    hdr = b"RIFF" + struct.pack("<I", 36 + 8 + f.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 2, 22050, 22050 * 8, 8, 32)
    junk = b"LIST" + struct.pack("<I", 3) + b"abc\0"                                   # odd-sized chunk is padded
    y, sr = fa.AudioWAV.read(hdr + junk + b"data" + struct.pack("<I", f.nbytes) + f.tobytes())
    assert sr == 22050 and np.array_equal(y, f)
    with pytest.raises(ValueError):
        fa.AudioWAV.read(b"RIFFxxxxWAVE")


def test_wav_reader_survives_damaged_files(fa):
    """The reader is a C function fed with file bytes: truncations, bit flips and random chunk sizes must end in a result or in
    ValueError, never in a crash or an out-of-bounds read (the copies below are exact-length numpy buffers).  A file that ends inside a
    frame yields its whole frames."""
    rng = np.random.default_rng(0)
    f = (rng.standard_normal((50, 2)) * 0.1).astype(np.float32)
    pcm = (rng.standard_normal(101) * 8000).astype("<i2")
    files = [b"RIFF" + struct.pack("<I", 36 + f.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 2, 22050, 22050 * 8, 8, 32) + b"data" + struct.pack("<I", f.nbytes) + f.tobytes(),
             b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()]
    y, _ = fa.AudioWAV.read(files[0][:-5])                      # 3 bytes short of the last frame + 2 more: 49 whole frames
    assert y.shape == (49, 2) and np.array_equal(y, f[:49])
    ok = bad = 0
    for it in range(3000):
        b = bytearray(files[it % 2])
        k = rng.integers(0, 4)
        if k == 0:
            b = b[:rng.integers(0, len(b))]
        elif k == 1:
            for _ in range(rng.integers(1, 6)):
                b[rng.integers(0, min(len(b), 48))] ^= 1 << rng.integers(0, 8)
        elif k == 2:
            pos = int(rng.choice([4, 16, 40]))
            b[pos:pos + 4] = struct.pack("<I", int(rng.choice([0, 1, 7, 2 ** 31 - 1, 2 ** 32 - 1, len(b), len(b) - 43])))
        else:
            b = b[:36] + bytes(rng.integers(0, 256, rng.integers(0, 40), dtype=np.uint8)) + b[36:]
        try:
            y, sr = fa.AudioWAV.read(bytes(b))
            assert y.ndim == 2 and y.shape[1] >= 1 and y.size * (4 if b[20] == 3 else 2) <= len(b)
            ok += 1
        except ValueError:
            bad += 1
    assert ok > 200 and bad > 200


@pytest.mark.gpu
@pytest.mark.parametrize("n,scale,normalize", [(1, 0.3, True), (1000, 0.2, True), (48000, 3.0, True), (48000, 3.0, False), (777, 0.0, True),
                                               (100003, 1e-3, True)])
def test_wav_writer_matches_restatement(fa, gpu_ctx, oracle_mod, n, scale, normalize):
    rng = np.random.default_rng(n)
    x = (scale * rng.standard_normal(n)).astype(np.float32)
    assert fa.AudioWAV.data(x, 24000, normalize, ctx=gpu_ctx) == oracle_mod.wav_pcm16(x, 24000, normalize)


@pytest.mark.gpu
def test_wav_writer_empty_and_header(fa, gpu_ctx, oracle_mod):
    data = fa.AudioWAV.data(np.zeros(0, np.float32), 16000, ctx=gpu_ctx)
    assert data == oracle_mod.wav_pcm16([], 16000) and len(data) == 44
    data = fa.AudioWAV.data(np.array([0.5, -0.25], np.float32), 44100.7, ctx=gpu_ctx)
    assert struct.unpack("<I", data[24:28])[0] == 44100 and struct.unpack("<I", data[28:32])[0] == 88201   # UInt32(sampleRate * 2)
    y, sr = fa.AudioWAV.read(data)
    assert sr == 44100 and np.array_equal(y[:, 0], np.array([32767, -16383], np.float32) / 32768.0)
