#!/bin/bash
# round 5, call 6: resampler rows kernel v2 (shared windows, DPP-broadcast taps): tests + probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q --timeout=600 -p no:cacheprovider -x ) > gpurun_out/r5/pytest6.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r5/pytest6.log | cut -c1-300
( time timeout 900 python scripts/r5/resample_probe.py ) 2>&1 | grep -v amdgpu.ids | cut -c1-300
