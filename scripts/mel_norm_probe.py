import sys, torch, numpy as np, ctypes as C
sys.path.insert(0, "/root/repo")
import fluidaudio_amd as fa
ctx = fa.default_context(0)
B, M, T = 1024, 128, 1501
x = torch.randn(B, M, T, device="cuda")
valid = torch.full((B,), T, dtype=torch.int32, device="cuda")
stream = torch.cuda.ExternalStream(ctx.stream)
L = fa.lib()
def run():
    ctx.check(L.fa_mel_normalize_per_feature_dev(ctx.handle, x.data_ptr(), B, M, T, T, valid.data_ptr()), "norm")
for _ in range(300): run()
ctx.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(100): run()
e1.record(stream); ctx.synchronize()
ms = e0.elapsed_time(e1) / 100
print("mel_norm ms", ms, "GB/s (read+write once)", 2 * x.numel() * 4 / ms / 1e6)
