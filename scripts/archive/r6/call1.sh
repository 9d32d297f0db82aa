#!/bin/bash
# round 6, call 1: Gram tiles that leave per-tile row minima (no second pass over the matrix): parity + start-up time
set -x
python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_e2e_digest.py -x -q -p no:cacheprovider 2>&1 | tail -n 5
python scripts/ahc_probe.py 43200,50000 --kinds mix --check 3000 2>&1 | tail -n 12
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_gram_prof -- python $GRAFT_REPO_ROOT/scripts/ahc_probe.py 43200 --kinds mix --check 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocprof_summary.py $(find gpurun_out/r06_gram_prof -name "*.db" | head -n 1) 2>&1 | head -n 25
