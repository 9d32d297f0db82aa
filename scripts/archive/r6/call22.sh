#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_ahc.py tests/test_gpu_ahc_adversarial.py tests/test_gpu_ahc_handover.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py -q -p no:cacheprovider > gpurun_out/r06_suite_dbg.txt 2>&1
tail -n 5 gpurun_out/r06_suite_dbg.txt
