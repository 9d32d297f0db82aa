#!/bin/bash
# round 6, call 11: fa_offline_cluster_batch_dev (inputs resident in HBM): parity test, K = 8 / 12 / 16 recordings of 8 h per call
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_pipeline.py -q -p no:cacheprovider 2>&1 | tail -n 3
python scripts/r6/batch_groups_probe.py 8,12,16 0 --dev 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_batch_dev.txt
