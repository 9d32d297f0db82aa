#!/bin/bash
# round 6, call 19: straight-line centroid pairs (+ odd-d tail): entries first (tree) / centroids first (cf), each with packed block records (p, cfp), against the last commit
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for rep in 1 2; do
for lib in libfluidaudio_hip_exp_old.so libfluidaudio_hip.so libfluidaudio_hip_exp_cf.so libfluidaudio_hip_exp_p.so libfluidaudio_hip_exp_cfp.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420
done
done
for lib in libfluidaudio_hip.so libfluidaudio_hip_exp_cf.so libfluidaudio_hip_exp_p.so libfluidaudio_hip_exp_cfp.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/r6/batch_groups_probe.py 8 0 --dev 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_round_diet2.txt
FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/libfluidaudio_hip_exp_cfp.so python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_ahc_handover.py tests/test_gpu_pipeline.py -q -x -p no:cacheprovider 2>&1 | tail -n 3
python -m pytest tests/test_gpu_ahc.py -q -x -p no:cacheprovider 2>&1 | tail -n 3
