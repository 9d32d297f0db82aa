"""A/B of the number of rows a wavefront of ctc_greedy_kernel keeps in flight: one library per value (built by scripts/archive/gpu_r4_call16.sh with
-DFA_CTC_ROWS=n, chosen through FLUIDAUDIO_HIP_LIBRARY), BASELINE configs[3] shapes ([1500, 1024] matrices), fp32 and fp16, HIP-event times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402

ctx = fa.default_context(0)
B, T, V = int(os.environ.get("FA_AB_BATCH", "4000")), 1500, int(os.environ.get("FA_AB_VOCAB", "1024"))
x32 = torch.randn((B, T, V), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda")
x32[:, :, V - 1] += 2.0
stream = torch.cuda.ExternalStream(ctx.stream)
ref = None
for name, x in (("f32", x32), ("f16", x32.half())):
    tok = torch.zeros((B, T), dtype=torch.int32, device="cuda")
    lens = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens, order=False)
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(3):
            fa.ctc_greedy_ids_dev(ctx, x, V - 1, tok, lens, order=False)
        e1.record(stream)
        ctx.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    gbs = x.numel() * x.element_size() / best / 1e6
    print(f"{os.environ.get('FLUIDAUDIO_HIP_LIBRARY', 'default').split('/')[-1]} {name} B={B} ms={best:.3f} GB/s={gbs:.0f} frac={gbs / 8000:.3f} "
          f"tokens={int(lens.sum())} checksum={int(tok.long().sum())}")
