#!/bin/bash
# round 5, call 5: TDT walk as a scalar state machine + register-resident rows (max + first-index search, buffer loads): tests and the leg at 1 024 .. 8 192 chunks
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 900 python -m pytest tests/test_gpu_tdt.py -m gpu -q --timeout=600 -p no:cacheprovider -x ) > gpurun_out/r5/pytest5.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r5/pytest5.log | cut -c1-300
( time timeout 1500 python scripts/tdt_leg_probe.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/tdt_leg_probe.log | cut -c1-400
