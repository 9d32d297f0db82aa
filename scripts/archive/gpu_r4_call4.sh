#!/bin/bash
# round 4, call 4: tiled VBx kernels, resampler with table rows in a VGPR + 16-byte stores, round counter in the hot state; single-chain probe; bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_resample.py tests/test_gpu_vbx.py tests/test_gpu_e2e_digest.py tests/test_gpu_ahc.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r4/pytest_call4.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r4/pytest_call4.log | cut -c1-900
( time timeout 300 python scripts/single_chain_probe.py ) > gpurun_out/r4/single_chain.log 2>&1; echo "single chain rc=$?"; grep '^{' gpurun_out/r4/single_chain.log | cut -c1-300
( time timeout 900 python bench.py ) > gpurun_out/r4/bench4.log 2> gpurun_out/r4/bench4.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench4.log > gpurun_out/r4/bench4.json; tail -5 gpurun_out/r4/bench4.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench4.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', j['roofline']['launch_period_us'])
for k, v in j['resample'].items():
    if isinstance(v, dict): print(k, v['ms_per_pass'], v['roofline']['frac'], v['within_2e-5'])
for k in ('e2e_8h_hard', 'e2e_8h_batch', 'ahc_50k', 'tdt'):
    print(k, json.dumps(j.get(k))[:1500])
PY
