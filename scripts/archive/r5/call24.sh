#!/bin/bash
# round 5, call 24 (final pass on the tree with the matrix-filtered reference-order run): PMC passes for the present bytes of ahc.hip (bench.py refuses
# stale traffic figures), the default bench line, the kernel trace of the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5 gpurun_out/summary
export TMPDIR=/tmp
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample
cp gpurun_out/summary/ahc_round_pmc.json profiles/r05_ahc_round_pmc.json   # bench.py reads profiles/
( time timeout 1200 python bench.py ) > gpurun_out/r5/bench24.log 2> gpurun_out/r5/bench24.err; echo "bench rc=$?"
tail -1 gpurun_out/r5/bench24.log > gpurun_out/r5/bench24.json; tail -4 gpurun_out/r5/bench24.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r5/rocprof_e2e24.log 2>&1; echo "rocprof e2e rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 14 | tee gpurun_out/summary/r05_e2e_kernel_stats.txt
rm -rf gpurun_out/prof_e2e
python - <<'PY'
import json
j = json.load(open('gpurun_out/r5/bench24.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', json.dumps(j['roofline'])[:600])
print('ties', json.dumps(j.get('ahc_ties'))[:900])
print('batch', json.dumps(j.get('e2e_8h_batch'))[:600])
PY
