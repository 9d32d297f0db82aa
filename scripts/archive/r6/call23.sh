#!/bin/bash
# round 6, call 23: cycle stamps of the round kernel (FA_AHC_PROFILE = 2: the overlapped timeline; 1: every stamp waits for memory)
cd "$GRAFT_REPO_ROOT" || exit 1
for m in 2 1; do
echo "## FA_AHC_PROFILE=$m"
FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_exp_prof$m.so python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r06_round_profile.txt
