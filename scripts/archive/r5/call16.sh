#!/bin/bash
# round 5, call 16: three uniform batches for 7 .. 10 long recordings — the whole GPU suite, the groups probe, the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r5/pytest16.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r5/pytest16.log | tail -2
( time timeout 1200 python bench.py ) > gpurun_out/r5/bench16.log 2> gpurun_out/r5/bench16.err; echo "bench rc=$?"
tail -1 gpurun_out/r5/bench16.log > gpurun_out/r5/bench16.json; tail -3 gpurun_out/r5/bench16.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r5/bench16.json').read())
print(j['value'], j['ms_per_step'])
print(json.dumps(j.get('e2e_8h_batch'))[:1500])
PY
