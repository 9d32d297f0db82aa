#!/bin/bash
# round 6, call 24: sizes + centroids requested by wave 0 alone (|ca - cb|^2 and the sizes handed over through LDS) against the tree
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_l.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 5400,43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_l.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 8 0 --dev 2>&1 | grep -v amdgpu.ids
done
echo "## bit-exactness, exp_l"
FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/libfluidaudio_hip_exp_l.so python scripts/ahc_probe.py 300,3000 --kinds iid,mix --check 5000 2>&1 | grep -v amdgpu.ids | cut -c1-120
} | tee gpurun_out/r06_round_leader.txt
