#!/bin/bash
# round 3: bench line, rocprof kernel traces (headline step, mel, ctc), PMC passes (ahc_round_t, ctc, mel)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > gpurun_out/r3/bench.log 2> gpurun_out/r3/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/r3/bench.log > gpurun_out/r3/bench.json; cut -c1-3000 gpurun_out/r3/bench.json; tail -5 gpurun_out/r3/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam ) > gpurun_out/r3/rocprof_e2e.log 2>&1; echo "rocprof e2e rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_mel" -o mel -- python "$GRAFT_REPO_ROOT/bench.py" --only-mel ) > gpurun_out/r3/rocprof_mel.log 2>&1; echo "rocprof mel rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 12 | tee gpurun_out/summary/e2e_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/prof_mel/mel_results.db --top 6 | tee gpurun_out/summary/mel_kernel_stats.txt
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam
bash scripts/gpu_pmc_kernel.sh ctc ctc_greedy "ctc.hip" python $GRAFT_REPO_ROOT/scripts/ctc_probe.py
bash scripts/gpu_mel_pmc.sh > gpurun_out/r3/pmc_mel.log 2>&1; echo "mel pmc rc=$?"
python scripts/pmc_summary.py mel_kernel $(find gpurun_out/pmc -name "*.db") > gpurun_out/summary/mel_pmc.json
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
rm -rf gpurun_out/prof_mel gpurun_out/prof_e2e; find gpurun_out/pmc -name "*.db" -delete
