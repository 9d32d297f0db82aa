#!/bin/bash
# round 5, call 4: TDT walk with the soft-max only on emission + the 4 096-chunk leg; uniform-batch graph cached in the context; the full bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 900 python -m pytest tests/test_gpu_tdt.py tests/test_gpu_ahc.py tests/test_gpu_pipeline.py -m gpu -q --timeout=600 -p no:cacheprovider -x ) > gpurun_out/r5/pytest4.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r5/pytest4.log | cut -c1-300
python scripts/batch_phases_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
( time timeout 1200 python bench.py ) > gpurun_out/r5/bench4.json 2> gpurun_out/r5/bench4.err; echo "bench rc=$?"
tail -3 gpurun_out/r5/bench4.err | cut -c1-300
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r5/bench4.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms/step", l["ms_per_step"])
for k in ("tdt","tdt_4096_fp16"):
    t=l.get(k,{}); print(k, {x:t.get(x) for x in ("ms_per_pass","ids_equal_table_walk_all_chunks","ids_equal_cpu_restatement_all_chunks","rows_per_chunk_mean","rows_of_the_longest_chunk","error")}, t.get("roofline",{}).get("frac"))
print("e2e_16x1h", l.get("e2e_16x1h"))
print({k:(v.get("audio_hours_per_s"),v.get("us_per_round")) for k,v in l.get("e2e_8h_batch",{}).items() if k.startswith("x")})
PY
