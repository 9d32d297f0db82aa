#!/bin/bash
# round 5, call 20: where a selection of the matrix-filtered reference-order run spends its time (FA_ROM_PROFILE build of ahc.hip, clock stamps per phase)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/scripts/ubench/libfluidaudio_hip_romprof.so ROM_PROBE_ONLY=1 timeout 300 python scripts/r5/rom_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/rom_profile20.log | cut -c1-700
