#!/bin/bash
# round 4, call 21: finer cycle stamps inside the selection of the beam walk
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
FA_BEAM_PROF=1 timeout 300 python scripts/beam_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/beam_probe21.txt
FA_BEAM_PROF=1 timeout 300 python scripts/beam_probe.py --batch 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe21.txt
