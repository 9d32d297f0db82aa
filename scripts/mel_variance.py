import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import fluidaudio_amd as fa
import bench
ctx = fa.default_context(0)
stream = torch.cuda.ExternalStream(ctx.stream)
B = 1024
d_pcm = bench.synth_pcm(torch, B, 1234)
offsets = np.arange(B + 1, dtype=np.int64) * bench.CHUNK_SAMPLES
mel = fa.AudioMelSpectrogram(ctx=ctx)
plan = mel.plan(offsets, layout="mel_major")
d_out = torch.empty(plan.out_shape(), dtype=torch.float32, device="cuda")
d_len = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(5): plan.execute(d_pcm, d_out, d_len)
ctx.synchronize()
n = 60
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record(stream)
for i in range(n):
    plan.execute(d_pcm, d_out, d_len); ev[i + 1].record(stream)
ctx.synchronize(); torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(" ".join(f"{m:.3f}" for m in ms))
print("mean", np.mean(ms), "median", np.median(ms), "min", np.min(ms), "max", np.max(ms))
