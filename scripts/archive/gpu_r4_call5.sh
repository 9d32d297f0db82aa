#!/bin/bash
# round 4, call 5: small-factor interpolation kernel, one-wavefront-per-chunk TDT walk, mel interior-tile requests (A/B), bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_resample.py tests/test_gpu_tdt.py tests/test_gpu_mel.py -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r4/pytest_call5.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r4/pytest_call5.log | cut -c1-900
for v in 0 1 0 1; do FA_MEL_V4_INTERIOR=$v timeout 300 python bench.py --only-mel 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())['mel']
print('FA_MEL_V4_INTERIOR=$v', 'kernel_ms_avg %.4f' % j['roofline']['kernel_ms_avg'], 'min %.4f' % j['roofline']['kernel_ms_min'], 'frac %.4f' % j['roofline']['frac'])
"; done | tee gpurun_out/summary/mel_interior_ab.txt
( time timeout 900 python bench.py ) > gpurun_out/r4/bench5.log 2> gpurun_out/r4/bench5.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench5.log > gpurun_out/r4/bench5.json; tail -5 gpurun_out/r4/bench5.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench5.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', j['roofline']['launch_period_us'])
for k, v in j['resample'].items():
    if isinstance(v, dict): print(k, v['ms_per_pass'], v['roofline']['frac'], v['within_2e-5'])
for k in ('tdt', 'mel'):
    print(k, json.dumps(j.get(k))[:1200])
PY
