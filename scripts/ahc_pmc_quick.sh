#!/bin/bash
# HBM traffic of ahc_round_t for the present bytes of ahc.hip: two PMC passes over ONE linkage of the 8 h bench session (the quick form of gpu_pmc_kernel.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_ahcq gpurun_out/summary
cat > /tmp/ahc_one.py <<'PY'
import os, sys
import numpy as np
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fluidaudio_amd as fa
from e2e_inputs import e2e_session
x = e2e_session(8.0, 12, seed=5)["emb"].astype(np.float64)
x /= np.sqrt((x * x).sum(axis=1, keepdims=True))
st, z = fa.linkage(x)
assert st == 0
PY
for c in "tcc1 FETCH_SIZE" "tcc2 WRITE_SIZE"; do set -- $c; ( cd /tmp && timeout 300 rocprofv3 --pmc $2 GRBM_GUI_ACTIVE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_ahcq/$1" -o $1 -- python /tmp/ahc_one.py ) > gpurun_out/pmc_ahcq/$1.log 2>&1; echo "ahcq/$1 rc=$?"; done
python scripts/pmc_summary.py ahc_round_t $(find gpurun_out/pmc_ahcq -name "*.db") > gpurun_out/summary/ahc_round_pmc.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
p = 'gpurun_out/summary/ahc_round_pmc.json'
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256(('ahc.hip',))
j['kernel_sources'] = ['ahc.hip']
j['workload'] = 'one fa_ahc_linkage of the 8 h bench session (43 200 x 256), scripts/ahc_pmc_quick.sh'
json.dump(j, open(p, 'w'), indent=1)
print({k: v for k, v in j.items() if k != 'counters'})
PY
find gpurun_out/pmc_ahcq -name "*.db" -delete
