#!/bin/bash
# round 5, call 21: the selection after its second latency pass (layered warm-up requests, single-hit fast path, pipelined sums): tests of both forms, the
# 8 h session probe, the phase profile (FA_ROM_PROFILE build), a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5 gpurun_out/summary
( time timeout 400 python -m pytest tests/test_gpu_ahc_adversarial.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "both_reference_order_forms or reference_order_mode or batch_with_tied" ) > gpurun_out/r5/pytest21.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r5/pytest21.log | cut -c1-600
FA_AHC_DEBUG=1 timeout 300 python scripts/r5/rom_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "^ahc" | tee gpurun_out/r5/rom_probe21.log | cut -c1-400
FLUIDAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/scripts/ubench/libfluidaudio_hip_romprof.so ROM_PROBE_ONLY=1 timeout 300 python scripts/r5/rom_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/rom_profile21.log | cut -c1-700
( cd /tmp && ROM_PROBE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_rom" -o rom -- python "$GRAFT_REPO_ROOT/scripts/r5/rom_probe.py" ) > gpurun_out/r5/rocprof_rom21.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_rom/rom_results.db --top 4 | tee gpurun_out/summary/r05_rom_kernel_stats.txt
rm -rf gpurun_out/prof_rom
