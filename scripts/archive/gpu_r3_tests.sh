#!/bin/bash
# round 3, call 1: the whole GPU suite (new: 8 h digests, adversarial AHC, workspace policy, torch.stft second opinion)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
( time timeout 1700 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider --deselect tests/test_gpu_workspace.py ) > gpurun_out/r3/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r3/pytest_gpu.log | cut -c1-400
( time timeout 600 python -m pytest tests/test_gpu_workspace.py -m gpu -q --timeout=300 -p no:cacheprovider ) > gpurun_out/r3/pytest_ws.log 2>&1; echo "pytest ws rc=$?"
tail -30 gpurun_out/r3/pytest_ws.log | cut -c1-400
