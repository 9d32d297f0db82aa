#!/bin/bash
# round 6, call 10: (a) rounds per graph replay 512 vs 2 048 = what the host's read-back between replays costs the chain;
# (b) the TDT walk with one line of every possible next row requested ahead (FA_TDT_WARM build)
cd "$GRAFT_REPO_ROOT" || exit 1
L=fluidaudio_amd/csrc
{
for lib in libfluidaudio_hip.so libfluidaudio_hip_rpg2048.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_rpg_probe.txt
{
for lib in libfluidaudio_hip.so libfluidaudio_hip_tdtwarm.so; do
  echo "## $lib"; FLUIDAUDIO_HIP_LIBRARY=$PWD/$L/$lib python scripts/tdt_leg_probe.py 1024:float32,4096:float32,1024:float16 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_tdt_warm_probe.txt
