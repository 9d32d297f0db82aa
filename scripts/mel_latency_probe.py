import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import fluidaudio_amd as fa
from fluidaudio_amd import _lib as L
import torch
ctx = fa.default_context()
rng = np.random.default_rng(11)
a = (rng.uniform(-1, 1, 160000) * 0.1).astype(np.float32)
mel = fa.AudioMelSpectrogram(ctx=ctx)
for _ in range(10): mel.compute_flat(a)
def bench(f, n=300):
    for _ in range(20): f()
    t=[]; 
    for _ in range(n):
        t0=time.perf_counter(); f(); t.append(time.perf_counter()-t0)
    t=np.sort(t); return 1e6*t[len(t)//2]
print("compute_flat p50 us", bench(lambda: mel.compute_flat(a)))
cfg = L.MelConfig(); L.lib().fa_mel_default_config(C.byref(cfg))
offs = np.array([0, 160000], np.int64)
def plan():
    p = C.c_void_p()
    L.lib().fa_mel_plan_create(ctx.handle, C.byref(cfg), offs.ctypes.data, 1, None, 0, C.byref(p))
    L.lib().fa_mel_plan_destroy(p)
print("plan create+destroy p50 us", bench(plan))
x = torch.zeros(160000, device='cuda'); 
def mallocs():
    for _ in range(3):
        t = torch.cuda.caching_allocator_alloc(1 << 20); torch.cuda.caching_allocator_delete(t)
h = torch.from_numpy(a)
def copies():
    x.copy_(h); y = torch.empty(128*1001, device='cuda'); y.cpu()
print("h2d 640KB + d2h 512KB (torch, pageable) p50 us", bench(copies))
