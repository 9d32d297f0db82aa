#!/usr/bin/env python3
"""Throughput probe for the device CTC beam search: B utterances of [T, V] log-probs, beam 100, 40 token candidates."""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--vocab", type=int, default=1025)
    ap.add_argument("--lm", type=int, default=1)
    a = ap.parse_args()
    import torch
    ctx = fa.default_context()
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(a.batch, a.frames, a.vocab, device="cuda", generator=g) * 3.0
    x[:, :, a.vocab - 1] += 4.0
    lp = torch.log_softmax(x, dim=-1).contiguous()
    words = ["the", "cat", "sat", "dog", "on", "mat", "a", "in", "of", "to"]
    voc = {v: ("\u2581" + words[v % len(words)] if v % 3 == 0 else "abcdefgh"[v % 8]) for v in range(a.vocab - 1)}
    arpa = "\\data\\\n\\1-grams:\n" + "".join(f"-{1 + 0.1 * i:.1f}\t{w}\t-0.3\n" for i, w in enumerate(words)) + "\\2-grams:\n" + \
        "".join(f"-0.{5 + i}\t{words[i]}\t{words[(i + 1) % len(words)]}\n" for i in range(len(words))) + "\\end\\\n"
    lm = fa.ARPALanguageModel(arpa, ctx=ctx) if a.lm else None
    vocab = fa.CtcVocabulary(voc, a.vocab, ctx)
    L = fa.lib()
    tok = torch.zeros(a.batch, a.frames, dtype=torch.int32, device="cuda")
    lens = torch.zeros(a.batch, dtype=torch.int32, device="cuda")
    sc = torch.zeros(a.batch, dtype=torch.float32, device="cuda")

    def run():
        ctx.check(L.fa_ctc_beam_search_batch_dev(ctx.handle, lp.data_ptr(), a.batch, a.frames, a.vocab, a.vocab, a.frames * a.vocab, None,
                                                 vocab.handle, lm.handle if lm else None, 100, 0.3, 0.0, a.vocab - 1, 40, tok.data_ptr(),
                                                 lens.data_ptr(), sc.data_ptr()), "beam")
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"beam_search": {"batch": a.batch, "frames": a.frames, "vocab": a.vocab, "lm": bool(a.lm), "seconds": dt,
                                      "utterances_per_s": a.batch / dt, "audio_hours_per_s": a.batch * a.frames * 0.01 / 3600 / dt,
                                      "us_per_frame_per_utterance_slot": dt / a.frames * 1e6, "mean_len": float(lens.float().mean()),
                                      "tokens_checksum": int((tok.long() * (torch.arange(a.frames, device="cuda") % 251 + 1)).sum()),
                                      "scores_sum": float(sc.double().sum())}}))


if __name__ == "__main__":
    main()
