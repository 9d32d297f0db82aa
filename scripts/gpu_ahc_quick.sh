#!/bin/bash
# quick look at the AHC round: bit-exactness at 4 000 points, timing of the 8 h session and of 50 000 points
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/exp
for lib in fluidaudio_amd/csrc/libfluidaudio_hip.so $(ls scripts/exp_lib_*.so 2>/dev/null); do
  echo "== $lib"
  FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python scripts/ahc_probe.py 4000 --kinds iid,mix --modes 0 --check 10000 2>&1 | grep -v amdgpu.ids | cut -c1-330
  FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python scripts/ahc_probe.py 43200 --kinds e2e --modes 0 --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-500
  FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 python scripts/ahc_probe.py 50000 --kinds iid --modes 0 --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-500
done 2>&1 | tee gpurun_out/exp/ahc_quick.log
