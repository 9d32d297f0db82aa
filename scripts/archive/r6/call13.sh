#!/bin/bash
# round 6, call 13: TDT rows in pieces of 1 / 2 / 4 / 8 logits per request and lane (AB build, FA_TDT_PIECE): parity on the release build, the leg per piece
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_tdt.py -q -p no:cacheprovider 2>&1 | tail -n 3
{
for p in 1 2 4; do
  echo "## fp32 piece $p"; FA_TDT_PIECE=$p FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_ab.so python scripts/tdt_leg_probe.py 1024:float32,4096:float32 2>&1 | grep -v amdgpu.ids
done
for p in 1 2 4 8; do
  echo "## fp16 piece $p"; FA_TDT_PIECE=$p FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_ab.so python scripts/tdt_leg_probe.py 1024:float16,4096:float16 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_tdt_piece_probe.txt
