#!/bin/bash
# round 4, call 12: per-context buffer cache + persistent worker contexts of the clustering stage: whole GPU suite, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r4/pytest_call12.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r4/pytest_call12.log | cut -c1-900
( time timeout 900 python bench.py ) > gpurun_out/r4/bench12.log 2> gpurun_out/r4/bench12.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench12.log > gpurun_out/r4/bench12.json; tail -3 gpurun_out/r4/bench12.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench12.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'stages', j['e2e_8h']['stages_s'])
print({k: (v['audio_hours_per_s'], v['wall_s']) for k, v in j['e2e_8h_batch'].items() if k.startswith('x')})
print('16x1h', j['e2e_16x1h'])
print('hard', j['e2e_8h_hard']['seconds_per_recording'], j['e2e_8h_hard']['stages_s'])
print('ahc_batch', j['ahc_batch']['batch_s'], 'mel single', j['mel_single_10s']['p50_ms'])
PY
