#!/bin/bash
# round 5, call 1: the new parity / degrade tests, then the whole bench line (baseline of the round; beam-search leg with events + repeats)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 900 python -m pytest tests/test_gpu_ctc.py tests/test_gpu_degrade.py tests/test_cabi_dropin.py tests/test_gpu_pipeline.py tests/test_gpu_workspace.py -m gpu -q --timeout=600 -p no:cacheprovider -x ) > gpurun_out/r5/pytest1.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r5/pytest1.log | cut -c1-400
( time timeout 900 python bench.py ) > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err; echo "bench rc=$?"
tail -3 gpurun_out/r5/bench1.err | cut -c1-300
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r5/bench1.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms/step", l["ms_per_step"])
print("beam", json.dumps(l.get("beam_search"))[:1500])
print("config", json.dumps(l["config"])[:3000])
PY
