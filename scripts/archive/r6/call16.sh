#!/bin/bash
# round 6, call 16: the round kernel of the tree (records per lane as a template count + centroids in 16-byte pieces) against: g = records requested only by the wavefronts
# that reduce them, h = hot state + NaN flag through wave 0 and LDS, f = matrix entries requested in front of sizes / centroids, and their combinations
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_g.so libfluidaudio_hip_exp_h.so libfluidaudio_hip_exp_f.so libfluidaudio_hip_exp_gh.so libfluidaudio_hip_exp_ghf.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
echo "## bit-exactness at 3 000 points (reference build on the host): release, ghf"
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_ghf.so; do
FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 300,3000 --kinds iid,mix --check 5000 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
} | tee gpurun_out/r06_round_requests_probe3.txt
python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py -q -x -p no:cacheprovider 2>&1 | tail -n 3
