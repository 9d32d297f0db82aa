#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
( time timeout 900 python -m pytest tests/test_gpu_ahc.py -m gpu -q --timeout=240 -p no:cacheprovider ) > gpurun_out/pytest_ahc.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_ahc.log
python - <<'PY' > gpurun_out/ahc_scaling.log 2>&1
import numpy as np, time, ctypes as C, torch, sys, json
sys.path.insert(0,'.')
import fluidaudio_amd as fa, oracle
from tests.conftest import speaker_mixture
ctx=fa.default_context(0)
def run(x,mode):
    n,d=x.shape; dx=torch.from_numpy(x).cuda(); dz=torch.zeros((n-1,4),dtype=torch.float64,device='cuda'); torch.cuda.synchronize()
    st=fa._lib.AhcStats(); t=time.perf_counter()
    rc=fa.lib().fa_ahc_linkage(ctx.handle,C.c_void_p(dx.data_ptr()),n,d,C.c_void_p(dz.data_ptr()),(n-1)*4,mode,1,C.byref(st))
    return rc,time.perf_counter()-t,st.as_dict(),dz.cpu().numpy()
for n in (2000,10000,20000,50000):
    for kind in ('iid','mix'):
        x = oracle.ahc_normalize(np.random.default_rng(0).standard_normal((n,256))) if kind=='iid' else speaker_mixture(n,256,64,0.02,0)
        for mode in (0,1):
            if mode==1 and n>20000: continue
            rc,t,s,z=run(x,mode); rc,t,s,z=run(x,mode)
            ok=None
            if n<=10000 and mode==0:
                t0=time.perf_counter(); sr,zr=oracle.linkage_ref(x); tref=time.perf_counter()-t0; ok=bool(np.array_equal(z,zr))
            else: tref=None
            print(json.dumps(dict(n=n,kind=kind,mode=mode,rc=rc,wall_s=round(t,4),ref_s=tref,bit_exact=ok,**{k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items()},us_per_round=round(1e3*s['merge_ms']/max(1,s['rounds']),2))),flush=True)
PY
cat gpurun_out/ahc_scaling.log
