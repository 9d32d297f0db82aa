#!/bin/bash
# round 6, call 18: the tree after the request diet of the round kernel (records per lane as a template count, centroids in 16-byte pieces, entries first):
# single chain against the last commit's library, the uniform batches (K = 8 / 12 resident), the linkage tests
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip_exp_old.so libfluidaudio_hip.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200,50000 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip_exp_old.so libfluidaudio_hip.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 8,12 0 --dev 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_round_diet.txt
python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_ahc_handover.py tests/test_gpu_pipeline.py -q -x -p no:cacheprovider 2>&1 | tail -n 3
