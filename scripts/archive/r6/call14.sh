#!/bin/bash
# round 6, call 14: the round kernel with fewer requests per round trip (three record pairs instead of four for <= 49 152 points; centroids as 16-byte pieces)
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_c3.so libfluidaudio_hip_exp_cpair.so libfluidaudio_hip_exp_c3cpair.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
} | tee gpurun_out/r06_round_requests_probe.txt
