#!/bin/bash
# round 5, call 9: start-ups of a uniform batch on two streams: tests + phases of the batch calls with / without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 1500 python -m pytest tests/test_gpu_ahc.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py tests/test_gpu_degrade.py -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r5/pytest9.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r5/pytest9.log | cut -c1-300
echo "--- two streams"; python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print({k: round(j[k],2) for k in ('count','wall_ms','linkage_phase_ms_max','ahc_init_ms','ahc_merge_ms','audio_hours_per_s')})
"
echo "--- one stream"; FA_AHC_NO_AUX_STREAM=1 python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print({k: round(j[k],2) for k in ('count','wall_ms','linkage_phase_ms_max','ahc_init_ms','ahc_merge_ms','audio_hours_per_s')})
"
