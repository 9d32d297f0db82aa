#!/bin/bash
# mel kernel variants: correctness of the default, then bench (mel only) of v3 / v4 x {3,4 workgroups per CU} x {ahead, not}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
( timeout 900 python -m pytest tests/test_gpu_mel.py -q --timeout=400 -p no:cacheprovider -x ) > gpurun_out/r2/pytest_mel.log 2>&1; echo "pytest mel rc=$?"; tail -5 gpurun_out/r2/pytest_mel.log
for cfg in "0 1" "3 1" "3 0" "4 1" "4 0"; do
  set -- $cfg
  echo "== FA_MEL_V4=$1 AHEAD=$2"
  FA_MEL_V4=$1 FA_MEL_V4_AHEAD=$2 timeout 300 python bench.py --skip-ahc --skip-ctc --skip-cpu --skip-e2e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f kernel_ms_avg %.4f min %.4f frac %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac']))"
done
