#!/bin/bash
# round 4, call 2: the uniform-layout batched round (tests, probe over K x register budget), the row-tiled resampler (tests), the bench line with
# the new legs (resample, tdt, ctc fp16, 8 h batch, hard session)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ahc.py tests/test_gpu_resample.py tests/test_gpu_workspace.py tests/test_gpu_pipeline.py tests/test_gpu_e2e_digest.py tests/test_gpu_tdt.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call2.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r4/pytest_call2.log | cut -c1-700
( time timeout 600 python scripts/uni_probe.py ) > gpurun_out/r4/uni_probe.log 2>&1; echo "uni probe rc=$?"; tail -45 gpurun_out/r4/uni_probe.log | cut -c1-400
( time timeout 900 python bench.py ) > gpurun_out/r4/bench2.log 2> gpurun_out/r4/bench2.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench2.log > gpurun_out/r4/bench2.json; tail -5 gpurun_out/r4/bench2.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench2.json'))
for k in ('value', 'ms_per_step', 'config'):
    print(k, json.dumps(j.get(k))[:1500])
for k in ('resample', 'tdt', 'ctc_fp16', 'e2e_8h_batch', 'e2e_8h_hard', 'ahc_batch', 'e2e_16x1h', 'e2e_8h_x4_in_flight'):
    print(k, json.dumps(j.get(k))[:1800])
PY
