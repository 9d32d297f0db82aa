#!/bin/bash
# round 4, call 13: two uniform batches side by side for eight or more medium recordings too: probe, linkage tests, quick PMC for the new ahc.hip bytes, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 300 python scripts/small_groups_probe.py ) > gpurun_out/r4/small_groups.log 2>&1; echo "probe rc=$?"; grep '^{' gpurun_out/r4/small_groups.log
( time timeout 900 python -m pytest tests/test_gpu_ahc.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call13.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4/pytest_call13.log | cut -c1-600
bash scripts/ahc_pmc_quick.sh
cp gpurun_out/summary/ahc_round_pmc.json profiles/r04_ahc_round_pmc.json
( time timeout 900 python bench.py ) > gpurun_out/r4/bench13.log 2> gpurun_out/r4/bench13.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench13.log > gpurun_out/r4/bench13.json; tail -3 gpurun_out/r4/bench13.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench13.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'traffic', j['roofline']['traffic'])
print({k: (v['audio_hours_per_s'], v['wall_s']) for k, v in j['e2e_8h_batch'].items() if k.startswith('x')})
print('16x1h', j['e2e_16x1h'])
print('ahc_batch', j['ahc_batch']['batch_s'])
PY
