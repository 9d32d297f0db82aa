#!/bin/bash
# round 6, call 17: control — the library of the last commit against the tree (and the tree with the plain 16-byte centroid read) on ONE box
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip_exp_old.so libfluidaudio_hip.so libfluidaudio_hip_exp_cs.so libfluidaudio_hip_exp_f.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
} | tee gpurun_out/r06_round_requests_probe4.txt
