#!/bin/bash
# round 6, call 20: the round kernel after the request diet (records per lane as a template count, packed block records, centroids in 16-byte pieces): full GPU suite,
# single chain + batches against the last commit's library
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
( time python -m pytest tests -q -m gpu -x -p no:cacheprovider ) 2>&1 | tail -n 6
{
for rep in 1 2; do
for lib in libfluidaudio_hip_exp_old.so libfluidaudio_hip.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 5400,43200,50000 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip_exp_old.so libfluidaudio_hip.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 4,8,12 0 --dev 2>&1 | grep -v amdgpu.ids
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | grep '"count": 16' | cut -c1-120
done
} | tee gpurun_out/r06_round_diet3.txt
