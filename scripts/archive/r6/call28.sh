#!/bin/bash
# round 6, call 28: the GPU suite on the poisoned-workspace build (the matrix of a halted AUTO attempt is not reused there: acquiring the workspace fills it)
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out/r6b
( time FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_poison.so python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6b/pytest_poison.log 2>&1; echo "pytest poison rc=$?"; tail -n 4 gpurun_out/r6b/pytest_poison.log | cut -c1-300
