#!/bin/bash
# round 4, call 17: ctc_greedy_kernel A/B: round-3 source, branch-free steps without / with nontemporal loads, nt + 2 / 4 rows in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_ctc.py -m gpu -q --timeout=300 -p no:cacheprovider ) > gpurun_out/r4/pytest_call17.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/r4/pytest_call17.log | cut -c1-300
: > gpurun_out/r4/ctc_ab2.txt
for rep in 1 2; do
for v in r3 nt0 default nt1rows2 nt1rows4; do
  lib=fluidaudio_amd/csrc/variants/libfa_ctc_$v.so
  [ $v = default ] && lib=fluidaudio_amd/csrc/libfluidaudio_hip.so
  FA_AB_BATCH=${FA_AB_BATCH:-6000} FLUIDAUDIO_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ctc_rows_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" >> gpurun_out/r4/ctc_ab2.txt
done
done
cat gpurun_out/r4/ctc_ab2.txt
