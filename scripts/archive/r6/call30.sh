#!/bin/bash
# round 6, call 30: the round's stores (new matrix row, row state, node / size / centroid / dendrogram row) issued BEHIND the block record (exp_ls) against the tree
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_ls.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 5400,43200,50000 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_ls.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 4,8,12 0 --dev 2>&1 | grep -v amdgpu.ids
done
echo "## bit-exactness, exp_ls"
FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/libfluidaudio_hip_exp_ls.so python scripts/ahc_probe.py 300,3000 --kinds iid,mix --check 5000 2>&1 | grep -v amdgpu.ids | cut -c1-120
} | tee gpurun_out/r06_round_late_stores.txt
FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/libfluidaudio_hip_exp_ls.so python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_ahc_handover.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py tests/test_gpu_degrade.py -q -p no:cacheprovider 2>&1 | tail -n 3
