#!/bin/bash
# round 4, call 23: beam walk without the profiling registers: tests, timing at batch 512 / 1024 / 2048, the 3-waves-per-SIMD build beside it, per-phase cycles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_beam.py -m gpu -q -x --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call23.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r4/pytest_call23.log | head -20 | cut -c1-400
: > gpurun_out/r4/beam_probe23.txt
for b in 512 1024 2048; do
  timeout 300 python scripts/beam_probe.py --batch $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe23.txt
  FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/variants/libfa_beam_w3.so timeout 300 python scripts/beam_probe.py --batch $b 2>&1 | grep -v amdgpu.ids | sed "s/^/w3 /" | tee -a gpurun_out/r4/beam_probe23.txt
done
FA_BEAM_PROF=1 timeout 300 python scripts/beam_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe23.txt
timeout 300 python scripts/beam_probe.py --batch 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe23.txt
