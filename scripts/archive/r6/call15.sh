#!/bin/bash
# round 6, call 15: fewer requests per round trip of the round kernel, step by step (a = three record pairs + centroids in 16-byte pieces; b = a + both sizes in one
# request; c = a + the hot state in one request; d = a + nontemporal row store; e = a + b + c)
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_a.so libfluidaudio_hip_exp_b.so libfluidaudio_hip_exp_c.so libfluidaudio_hip_exp_d.so libfluidaudio_hip_exp_e.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
echo "## bit-exactness of e at 3 000 points (reference build on the host)"
FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/libfluidaudio_hip_exp_e.so python scripts/ahc_probe.py 3000 --kinds iid,mix --check 5000 2>&1 | grep -v amdgpu.ids | cut -c1-200
} | tee gpurun_out/r06_round_requests_probe2.txt
