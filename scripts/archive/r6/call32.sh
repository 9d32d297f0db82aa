#!/bin/bash
# round 6, call 32: launches over several problems at ONE slot per thread (K = 2 / 3 long recordings, many short ones): matrix entries behind their branch (eb1) or unconditional (tree)
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_eb1.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 2,3 0 --dev 2>&1 | grep -v amdgpu.ids
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | grep '"count": 16' | cut -c1-110
done
done
} | tee gpurun_out/r06_round_eb1.txt
