#!/bin/bash
# one gpurun call: smoke, gpu tests, bench, rocprof kernel traces (mel, ahc), PMC passes for the mel kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/smoke.log
( time timeout 1200 python -m pytest tests -m gpu -q --timeout=240 -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -4 gpurun_out/bench.log | cut -c1-3000
export TMPDIR=/tmp
rm -rf gpurun_out/prof_mel gpurun_out/prof_ahc gpurun_out/pmc
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_mel" -o mel -- python "$GRAFT_REPO_ROOT/bench.py" --skip-ahc --skip-ctc --skip-cpu --skip-e2e ) > gpurun_out/rocprof_mel.log 2>&1; echo "rocprof mel rc=$?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ahc" -o ahc -- python "$GRAFT_REPO_ROOT/scripts/ahc_probe.py" 50000 --kinds iid --modes 0 --check 0 ) > gpurun_out/rocprof_ahc.log 2>&1; echo "rocprof ahc rc=$?"
timeout 300 python scripts/ahc_probe.py 2000,10000,20000,50000 --modes 0 2>&1 | grep -v amdgpu.ids > gpurun_out/ahc_scaling.log
timeout 300 python scripts/beam_probe.py --batch 512 2>&1 | grep -v amdgpu.ids > gpurun_out/beam_probe.log; tail -1 gpurun_out/beam_probe.log
FA_MEL_PROF=1 python bench.py --skip-ahc --skip-ctc --skip-cpu --skip-e2e 2>&1 | grep "mel profile" | tail -1 > gpurun_out/summary_mel_tile_profile.txt; cat gpurun_out/summary_mel_tile_profile.txt
bash scripts/gpu_mel_pmc.sh > gpurun_out/pmc.log 2>&1; echo "pmc rc=$?"
mkdir -p gpurun_out/summary
python scripts/rocprof_summary.py gpurun_out/prof_mel/mel_results.db --top 8 | tee gpurun_out/summary/mel_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/prof_ahc/ahc_results.db --top 8 | tee gpurun_out/summary/ahc_kernel_stats.txt
python scripts/pmc_summary.py mel_kernel gpurun_out/pmc/*/*.db > gpurun_out/summary/mel_pmc.json
grep -v '"counters"' gpurun_out/summary/mel_pmc.json | tail -14
# the rocpd databases are tens of MB each; only the summaries travel back
rm -rf gpurun_out/prof_mel gpurun_out/prof_ahc gpurun_out/pmc/*/
