#!/bin/bash
# round 5, call 10: reference-order scans with 64 coordinates in flight: adversarial parity + what a tie costs at 43 200 x 256
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 1500 python -m pytest tests/test_gpu_ahc_adversarial.py tests/test_gpu_workspace.py -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r5/pytest10.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r5/pytest10.log | cut -c1-300
python scripts/r5/ties_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/ties_probe.log
