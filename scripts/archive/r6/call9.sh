#!/bin/bash
# round 6, call 9: batches side by side by K on the whole call; idle time between replays of the round graph
cd "$GRAFT_REPO_ROOT" || exit 1
python scripts/r6/batch_groups_probe.py 8,12 2,3,4 2>&1 | tee gpurun_out/r06_batch_groups.txt
bash scripts/r6/gap_probe.sh 2>&1 | tee gpurun_out/r06_gap_probe.txt
