#!/bin/bash
# round 5, call 19: as call 18 after the symmetric matrix (mirror workgroups inside the selection launch), eight-level heap blocks, parity-buffered state
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5 gpurun_out/summary
( time timeout 400 python -m pytest tests/test_gpu_ahc_adversarial.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "both_reference_order_forms or reference_order_mode or batch_with_tied" ) > gpurun_out/r5/pytest19.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r5/pytest19.log | cut -c1-600
FA_AHC_DEBUG=1 timeout 300 python scripts/r5/rom_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "^ahc: N" | tee gpurun_out/r5/rom_probe19.log | cut -c1-400
( cd /tmp && ROM_PROBE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_rom" -o rom -- python "$GRAFT_REPO_ROOT/scripts/r5/rom_probe.py" ) > gpurun_out/r5/rocprof_rom19.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_rom/rom_results.db --top 10 | tee gpurun_out/summary/r05_rom_kernel_stats.txt
rm -rf gpurun_out/prof_rom
