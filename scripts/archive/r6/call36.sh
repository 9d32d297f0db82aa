#!/bin/bash
# round 6, call 36: the Gram start-up launched over the tiles on and below the diagonal only (exp_tri) against the square grid whose upper half returns at once
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2 3; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_tri.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200,50000 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-40,150-330
done
done
echo "## exactness exp_tri"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/libfluidaudio_hip_exp_tri.so python scripts/ahc_probe.py 300,3000 --kinds iid,mix --check 5000 2>&1 | grep -v amdgpu.ids | cut -c1-120
} | tee gpurun_out/r06_gram_tri_grid.txt
