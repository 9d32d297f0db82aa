#!/bin/bash
# round 6, call 21: the GPU suite with its failure text
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r06_suite_dbg.txt 2>&1
tail -n 5 gpurun_out/r06_suite_dbg.txt
