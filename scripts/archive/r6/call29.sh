#!/bin/bash
# round 6, call 29: the centroid sums without selects (rows_finite) and with the denominator on its own wavefront: parity, the stage's phases, K = 8 / 12
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_post.py tests/test_gpu_pipeline.py tests/test_gpu_e2e_digest.py tests/test_gpu_vbx.py -q -p no:cacheprovider 2>&1 | tail -n 3
python scripts/r6/batch_groups_probe.py 8,12 0 --dev 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_centroid_fast.txt
python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a gpurun_out/r06_centroid_fast.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_centroid_fast.txt
import sys, os, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import fluidaudio_amd as fa
from e2e_inputs import e2e_session
s = e2e_session(8.0, 12, seed=5)
ctx = fa.default_context()
for rep in range(3):
    r = fa.cluster_embeddings(s["emb"], s["rho"], s["chunks"], s["phi"], ctx=ctx)
print(json.dumps({"single_8h": r.timings}))
PY
