#!/bin/bash
# mel: buffer-load fetch (this tree) — tests, then the mel leg of the bench three times (DEEP on = default, off once)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
( timeout 900 python -m pytest tests/test_gpu_mel.py -q --timeout=400 -p no:cacheprovider -x ) > gpurun_out/r2/pytest_mel.log 2>&1; echo "pytest mel rc=$?"; tail -3 gpurun_out/r2/pytest_mel.log
run() { echo "== $*"; env "$@" timeout 300 python bench.py --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam 2>&1 | grep -E "profile|metric|mean by" | sed 's/.*"ms_per_step": \([0-9.]*\).*"kernel_ms_avg": \([0-9.]*\).*/ms_per_step \1 kernel_ms_avg \2/' | tail -7; }
run FA_X=1
run FA_MEL_V4_DEEP=0
run FA_X=1
run FA_MEL_PROF=1
