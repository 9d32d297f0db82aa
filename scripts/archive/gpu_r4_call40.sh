#!/bin/bash
# round 4, call 40: the whole GPU suite on the tree as committed
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r4/pytest_gpu40.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4/pytest_gpu40.log | cut -c1-600
