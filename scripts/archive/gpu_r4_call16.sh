#!/bin/bash
# round 4, call 16: ctc_greedy_kernel with several rows in flight per wavefront: parity tests, then the A/B over rows-at-once (1 = the
# round-3 schedule, 2, 4, 8; variant libraries built beforehand with -DFA_CTC_ROWS=n under fluidaudio_amd/csrc/variants/)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_ctc.py -m gpu -q --timeout=300 -p no:cacheprovider ) > gpurun_out/r4/pytest_call16.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4/pytest_call16.log | cut -c1-400
: > gpurun_out/r4/ctc_rows_ab.txt
for n in 1 2 4 8; do
  lib=fluidaudio_amd/csrc/variants/libfa_ctc_rows$n.so
  [ $n = 4 ] && lib=fluidaudio_amd/csrc/libfluidaudio_hip.so
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ctc_rows_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/rows=$n /" >> gpurun_out/r4/ctc_rows_ab.txt
done
cat gpurun_out/r4/ctc_rows_ab.txt
