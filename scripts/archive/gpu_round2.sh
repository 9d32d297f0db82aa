#!/bin/bash
# round 2, one gpurun call: smoke, gpu tests, bench, rocprof kernel traces (mel, ctc, ahc), PMC passes for the mel kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2 gpurun_out/summary
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/r2/gpu.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r2/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r2/smoke.log | head -1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider ) > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r2/pytest_gpu.log
( time timeout 900 python bench.py ) > gpurun_out/r2/bench.log 2>&1; echo "bench rc=$?"
tail -4 gpurun_out/r2/bench.log | cut -c1-6000
export TMPDIR=/tmp
rm -rf gpurun_out/prof_mel gpurun_out/prof_ahc gpurun_out/prof_ctc gpurun_out/pmc
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_mel" -o mel -- python "$GRAFT_REPO_ROOT/bench.py" --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam ) > gpurun_out/r2/rocprof_mel.log 2>&1; echo "rocprof mel rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ctc" -o ctc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --clock-warm-s 0 --skip-ahc --skip-cpu --skip-e2e --skip-beam ) > gpurun_out/r2/rocprof_ctc.log 2>&1; echo "rocprof ctc rc=$?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ahc" -o ahc -- python "$GRAFT_REPO_ROOT/scripts/ahc_probe.py" 50000 --kinds iid --modes 0 --check 0 ) > gpurun_out/r2/rocprof_ahc.log 2>&1; echo "rocprof ahc rc=$?"
bash scripts/gpu_mel_pmc.sh > gpurun_out/r2/pmc.log 2>&1; echo "pmc rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_mel/mel_results.db --top 8 | tee gpurun_out/summary/mel_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/prof_ctc/ctc_results.db --top 8 | tee gpurun_out/summary/ctc_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/prof_ahc/ahc_results.db --top 8 | tee gpurun_out/summary/ahc_kernel_stats.txt
python scripts/pmc_summary.py mel_kernel gpurun_out/pmc/*/*.db > gpurun_out/summary/mel_pmc.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
j = json.load(open('gpurun_out/summary/mel_pmc.json'))
j['kernel_sources_sha256'] = bench.mel_kernel_sources_sha256()
j['kernel_sources'] = list(bench.MEL_KERNEL_SOURCES)
json.dump(j, open('gpurun_out/summary/mel_pmc.json', 'w'), indent=1)
print({k: v for k, v in j.items() if k != 'counters'})
PY
# the rocpd databases are tens of MB each; only the summaries travel back
rm -rf gpurun_out/prof_mel gpurun_out/prof_ahc gpurun_out/prof_ctc gpurun_out/pmc/*/
