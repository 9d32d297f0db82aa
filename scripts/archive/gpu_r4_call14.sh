#!/bin/bash
# round 4, call 14: host threads of the linkage batches catch everything (ahc.hip host side): linkage tests, quick PMC for the new bytes, short-recording
# latency probes on the final tree, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ahc.py tests/test_gpu_workspace.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call14.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4/pytest_call14.log | cut -c1-600
bash scripts/ahc_pmc_quick.sh
cp gpurun_out/summary/ahc_round_pmc.json profiles/r04_ahc_round_pmc.json
( timeout 300 python scripts/cluster_small_probe.py ) > gpurun_out/r4/cluster_small.log 2>&1; tail -1 gpurun_out/r4/cluster_small.log > gpurun_out/summary/cluster_small.json; grep "minutes" gpurun_out/r4/cluster_small.log | head -5 | cut -c1-300
( timeout 300 python scripts/ahc_small_probe.py ) > gpurun_out/r4/ahc_small.log 2>&1; tail -1 gpurun_out/r4/ahc_small.log > gpurun_out/summary/ahc_small.json; tail -3 gpurun_out/r4/ahc_small.log | cut -c1-700
( time timeout 900 python bench.py ) > gpurun_out/r4/bench14.log 2> gpurun_out/r4/bench14.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench14.log > gpurun_out/r4/bench14.json; tail -3 gpurun_out/r4/bench14.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench14.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'traffic', j['roofline']['traffic'])
print({k: (v['audio_hours_per_s'], v['wall_s']) for k, v in j['e2e_8h_batch'].items() if k.startswith('x')}, j['e2e_16x1h']['audio_hours_per_s'])
PY
