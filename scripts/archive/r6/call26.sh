#!/bin/bash
# round 6, call 26: which part of the unconditional-request change costs the single chain: b = matrix entries of one slot per thread behind their branch again,
# e = speculative block / centroid reads conditional again, c = both (= the last commit + the consumed reloads)
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip_exp_head.so libfluidaudio_hip.so libfluidaudio_hip_exp_b.so libfluidaudio_hip_exp_c.so libfluidaudio_hip_exp_e.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip_exp_b.so libfluidaudio_hip_exp_c.so libfluidaudio_hip_exp_e.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 8 0 --dev 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_round_uncond2.txt
