#!/bin/bash
# round 4, call 39: centroid sums with one add per row on the dependent chain (+0.0 for skipped rows, next rows' LDS reads ahead): parity tests, the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_pipeline.py tests/test_gpu_e2e_digest.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call39.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" gpurun_out/r4/pytest_call39.log | cut -c1-300
( timeout 600 python bench.py --steps 6 --warmup 2 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r4/bench39.log 2> gpurun_out/r4/bench39.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench39.log > gpurun_out/r4/bench39.json
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench39.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'digest', j['e2e_equals_reference_digest'])
print(j['e2e_8h']['stages_s'])
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r4/rocprof_e2e39.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 12 | grep -i "centroid\|scores\|hungarian\|vbx" 
rm -rf gpurun_out/prof_e2e
