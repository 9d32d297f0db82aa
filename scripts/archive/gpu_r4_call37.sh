#!/bin/bash
# round 4, call 37: the log-softmax row kernel on V = 1024 (16-byte path) and V = 1025 (4-byte path)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 300 python scripts/lsm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/lsm_probe.txt
