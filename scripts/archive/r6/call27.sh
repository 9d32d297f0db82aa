#!/bin/bash
# round 6, call 27: request forms per kernel family (single chain: behind their conditions; launches over several problems: unconditional) against the last commit
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip_exp_head.so libfluidaudio_hip.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 5400,43200,50000 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip_exp_head.so libfluidaudio_hip.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 2,4,8,12 0 --dev 2>&1 | grep -v amdgpu.ids
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | grep '"count": 16' | cut -c1-60
done
} | tee gpurun_out/r06_round_uncond3.txt
python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_ahc_handover.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py -q -p no:cacheprovider 2>&1 | tail -n 3
