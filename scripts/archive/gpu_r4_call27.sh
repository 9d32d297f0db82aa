#!/bin/bash
# round 4, call 27: beam-search tests incl. the batch that needs two launches (trie allocation cap)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( time timeout 900 python -m pytest tests/test_beam.py -m gpu -q -x --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call27.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4/pytest_call27.log | cut -c1-600
