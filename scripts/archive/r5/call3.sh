#!/bin/bash
# round 5, call 3: the linkage tests on the tree with 1 / 2 / 4 slots per thread (policy: 2 for launches of many workgroups and for 257..512-point recordings)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 1700 python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_workspace.py tests/test_gpu_pipeline.py tests/test_gpu_e2e_digest.py -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r5/pytest3.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r5/pytest3.log | cut -c1-400
