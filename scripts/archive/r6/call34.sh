#!/bin/bash
# round 6, call 34: the TDT walk with 2 / 4 wavefronts per chunk (AB build, FA_TDT_WPC): parity, then the leg at 1 024 / 2 048 / 4 096 chunks
cd "$GRAFT_REPO_ROOT" || exit 1
AB=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_ab.so
for w in 2 4; do echo "## parity WPC=$w"; FA_TDT_WPC=$w FLUIDAUDIO_HIP_LIBRARY=$AB python -m pytest tests/test_gpu_tdt.py -q -p no:cacheprovider 2>&1 | tail -n 2; done
{
for w in 1 2 4; do
  echo "## WPC $w"; FA_TDT_WPC=$w FLUIDAUDIO_HIP_LIBRARY=$AB python scripts/tdt_leg_probe.py 1024:float32,2048:float32,4096:float32,1024:float16,4096:float16 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_tdt_wpc_probe.txt
