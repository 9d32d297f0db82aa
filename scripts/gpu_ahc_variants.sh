#!/bin/bash
# AHC round-kernel experiments: every scripts/exp_lib_*.so (variants built on the build host) next to the product library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/exp
for lib in fluidaudio_amd/csrc/libfluidaudio_hip.so scripts/exp_lib_*.so; do
  echo "== $lib"
  FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib FA_AHC_DEBUG=1 timeout 300 python scripts/ahc_probe.py 4000 --kinds iid,mix --modes 0 --check 10000 2>&1 | grep -v amdgpu.ids | cut -c1-330
  FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib FA_AHC_DEBUG=1 timeout 300 python scripts/ahc_probe.py 43200 --kinds e2e --modes 0 --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-500
  FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib FA_AHC_DEBUG=1 timeout 300 python scripts/ahc_probe.py 50000 --kinds iid --modes 0 --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-500
done 2>&1 | tee gpurun_out/exp/ahc_variants.log
