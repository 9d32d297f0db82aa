#!/usr/bin/env python3
"""Log-softmax row kernel (CtcDecoder.swift:15-36 / CtcKeywordSpotter+Inference.swift:350-431): [B, 1500, V] fp32 -> log-probabilities, V = 1024 (16-byte path)
and V = 1025 (the model's own row length), HIP events, bytes = one read + one write of the matrix."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context(0)
stream = torch.cuda.ExternalStream(ctx.stream)
B, T = 2500, 1500
for V in (1024, 1025):
    x = torch.randn((B, T, V), device="cuda", dtype=torch.float32)
    o = torch.empty_like(x)
    fa.ctc_log_probs_dev(ctx, x, 1.0, 0.0, V - 1, d_out=o, order=False)
    ctx.synchronize()
    ok = bool(torch.allclose(o[:64], torch.log_softmax(x[:64], dim=-1), rtol=0, atol=2e-5))
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(3):
            fa.ctc_log_probs_dev(ctx, x, 1.0, 0.0, V - 1, d_out=o, order=False)
        e1.record(stream)
        ctx.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    gbs = 2 * x.numel() * 4 / best / 1e6
    print(json.dumps({"V": V, "ms": round(best, 3), "GB/s": round(gbs), "frac": round(gbs / 8000, 3), "close_to_torch": ok}), flush=True)
    del x, o
