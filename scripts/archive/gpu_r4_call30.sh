#!/bin/bash
# round 4, call 31 (one kernel build per path, plain loops, clamped loads): CTC greedy on rows of any alignment (Parakeet CTC: 1 025 logits per frame): parity tests, A/B against the previous library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( time timeout 600 python -m pytest tests/test_gpu_ctc.py -m gpu -q --timeout=300 -p no:cacheprovider ) > gpurun_out/r4/pytest_call31.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r4/pytest_call31.log | head -20 | cut -c1-400
: > gpurun_out/r4/ctc_unaligned_ab4.txt
for vocab in 1025 1024; do
for v in head default; do
  lib=fluidaudio_amd/csrc/variants/libfa_ctc_$v.so
  [ $v = default ] && lib=fluidaudio_amd/csrc/libfluidaudio_hip.so
  FA_AB_VOCAB=$vocab FA_AB_BATCH=6000 FLUIDAUDIO_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ctc_rows_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/V=$vocab $v /" | tee -a gpurun_out/r4/ctc_unaligned_ab4.txt
done
done
