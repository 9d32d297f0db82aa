#!/bin/bash
# round 4, call 38: log-softmax row kernel (16-byte path): requests first with clamped indices + streaming loads (default), the same + streaming stores (ntst), previous commit (head)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( timeout 600 python -m pytest tests/test_gpu_ctc.py -m gpu -q --timeout=300 -p no:cacheprovider ) > gpurun_out/r4/pytest_call38.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r4/pytest_call38.log | cut -c1-200
: > gpurun_out/r4/lsm_ab.txt
for v in head default ntst head default ntst; do
  lib=fluidaudio_amd/csrc/variants/libfa_ctc_$v.so
  [ $v = default ] && lib=fluidaudio_amd/csrc/libfluidaudio_hip.so
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/lsm_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a gpurun_out/r4/lsm_ab.txt
done
