"""One launch of the greedy CTC kernel on BASELINE configs[3] (10 000 x [1500, 1024] fp32) for the PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context(0)
B, T, V = 10000, 1500, 1024
x = torch.randn((B, T, V), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda")
x[:, :, V - 1] += 2.0
tok = torch.zeros((B, T), dtype=torch.int32, device="cuda")
lens = torch.zeros(B, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens, order=False)
ctx.synchronize()
print("tokens", float(lens.float().mean()))
