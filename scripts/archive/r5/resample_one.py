import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import fluidaudio_amd as fa
rate, up, down = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = fa.default_context()
n = rate * 3600
x = torch.randn(n, device="cuda", dtype=torch.float32) * 0.1
n_out = fa.lib().fa_resample_poly_frames(n, up, down)
y = torch.empty(n_out, device="cuda", dtype=torch.float32)
got = C.c_int64()
for _ in range(3):
    ctx.check(fa.lib().fa_resample_poly_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, up, down, C.c_void_p(y.data_ptr()), n_out, C.byref(got)), "resample")
ctx.synchronize()
