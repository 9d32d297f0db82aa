#!/bin/bash
# round 4, call 3: whole GPU suite (round kernels without the many-record path / cold-state loads, matrix-free fallback, batched staging of the
# row-tiled resampler), the uniform-batch probe, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r4/pytest_call3.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r4/pytest_call3.log | cut -c1-700
( time timeout 600 python scripts/uni_probe.py ) > gpurun_out/r4/uni_probe.log 2>&1; echo "uni probe rc=$?"; grep '^{' gpurun_out/r4/uni_probe.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['n'], r['K'], r['env'], 'us/round %.2f' % r['us_per_round'], 'wall %.3f' % r['wall_s'], 'h/s %.1f' % r.get('audio_hours_per_s_linkage_only', 0), r['equal_single'])
"; tail -3 gpurun_out/r4/uni_probe.log | cut -c1-300
( time timeout 900 python bench.py ) > gpurun_out/r4/bench3.log 2> gpurun_out/r4/bench3.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench3.log > gpurun_out/r4/bench3.json; tail -5 gpurun_out/r4/bench3.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench3.json'))
for k in ('value', 'ms_per_step', 'config', 'roofline'):
    print(k, json.dumps(j.get(k))[:1600])
for k in ('resample', 'e2e_8h_batch', 'e2e_8h_hard', 'ahc_batch', 'e2e_16x1h', 'e2e_8h_x4_in_flight', 'ahc_50k'):
    print(k, json.dumps(j.get(k))[:2500])
PY
