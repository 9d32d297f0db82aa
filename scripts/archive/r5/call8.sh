#!/bin/bash
# round 5, call 8: the whole GPU suite on the tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 -p no:cacheprovider ) > gpurun_out/r5/pytest8.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r5/pytest8.log | cut -c1-400
