#!/bin/bash
# round 4, call 25: cache policy of the resampler's loads / stores (default, nontemporal loads, nontemporal stores, both), two rounds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
: > gpurun_out/r4/resample_nt_ab.txt
for rep in 1 2; do
for v in default ld st ldst; do
  lib=fluidaudio_amd/csrc/variants/libfa_rs_$v.so
  [ $v = default ] && lib=fluidaudio_amd/csrc/libfluidaudio_hip.so
  FLUIDAUDIO_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/resample_nt_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/resample_nt_ab.txt
done
done
