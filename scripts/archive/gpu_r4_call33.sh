#!/bin/bash
# round 4, call 33: TDT decision without per-lane branches (selects, max-first soft-max reduction, clamped requests): tests, A/B against the previous library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( time timeout 600 python -m pytest tests/test_gpu_tdt.py -m gpu -q --timeout=300 -p no:cacheprovider ) > gpurun_out/r4/pytest_call33.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r4/pytest_call33.log | head -10 | cut -c1-400
: > gpurun_out/r4/tdt_ab.txt
for v in head default head default; do
  lib=fluidaudio_amd/csrc/variants/libfa_tdt_$v.so
  [ $v = default ] && lib=fluidaudio_amd/csrc/libfluidaudio_hip.so
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/tdt_leg_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/tdt_ab.txt
done
