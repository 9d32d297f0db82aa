#!/bin/bash
# round 4, call 24: beam walk with the blank's log-prob in the top-token table: tests, timing (previous library beside it), per-phase cycles, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_beam.py -m gpu -q -x --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call24.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r4/pytest_call24.log | head -20 | cut -c1-400
timeout 300 python scripts/beam_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/beam_probe24.txt
FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/variants/libfa_beam_prev.so timeout 300 python scripts/beam_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/prev /" | tee -a gpurun_out/r4/beam_probe24.txt
FA_BEAM_PROF=1 timeout 300 python scripts/beam_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe24.txt
timeout 300 python scripts/beam_probe.py --batch 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe24.txt
timeout 300 python scripts/beam_probe.py --lm 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/beam_probe24.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4/prof_beam -o beam -- python $GRAFT_REPO_ROOT/scripts/beam_probe.py ) > gpurun_out/r4/rocprof_beam.log 2>&1
python scripts/rocprof_summary.py $(find gpurun_out/r4/prof_beam -name "*.db" | head -1) --top 8 2>&1 | head -8 | tee gpurun_out/r4/beam_kernel_stats24.txt
find gpurun_out/r4/prof_beam -name "*.db" -delete
