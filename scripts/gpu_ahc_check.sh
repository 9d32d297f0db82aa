#!/bin/bash
# AHC-only GPU check: parity tests, scaling probe, kernel trace of a mid-size run, cycle-stamp profile build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_ahc.py -m gpu -q -x --timeout=240 -p no:cacheprovider ) > gpurun_out/pytest_ahc.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_ahc.log
timeout 600 python scripts/ahc_probe.py 2000,10000,20000 --modes 0,1 2>&1 | grep -v amdgpu.ids > gpurun_out/ahc_scaling.log
timeout 300 python scripts/ahc_probe.py 50000 --modes 0 --check 0 2>&1 | grep -v amdgpu.ids >> gpurun_out/ahc_scaling.log
cat gpurun_out/ahc_scaling.log
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ahc" -o ahc -- python "$GRAFT_REPO_ROOT/scripts/ahc_probe.py" 20000 --kinds iid --modes 0 --check 0 ) > gpurun_out/rocprof_ahc.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_ahc/ahc_results.db 2>&1 | head -8
# cycle-stamp build (diagnostic only; the shipped .so is rebuilt without the flag afterwards)
( cd fluidaudio_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DFA_AHC_PROFILE=1 -c ahc.hip -o ahc.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfluidaudio_hip.so ctx.o mel.o ctc.o tdt.o ahc.o vbx.o post.o resample.o ) && timeout 300 python scripts/ahc_probe.py 50000 --kinds iid,mix --modes 0 --check 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ahc_cycles.log
