#!/bin/bash
# round 4, call 8: rows kernel with the XCD-aware (tile, group) map — tests, PMC traffic of the resamplers, the bench line on the runtime's default
# hardware queues
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_pipeline.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call8.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r4/pytest_call8.log | cut -c1-700
name=r4k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=resample timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
summ() {  # <output stem> <kernel pattern> <source file>
  python scripts/pmc_summary.py "$2" $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/$1_pmc.json
  python - "$1" "$3" <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
p = f'gpurun_out/summary/{sys.argv[1]}_pmc.json'
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256((sys.argv[2],))
j['kernel_sources'] = [sys.argv[2]]
json.dump(j, open(p, 'w'), indent=1)
print(sys.argv[1], {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256')})
PY
}
summ resample_44100 poly_rows_kernelILi16 resample.hip
summ resample_22050 poly_rows_kernelILi8 resample.hip
summ resample_8000 poly_interp_kernel resample.hip
summ resample_48000 poly_decim_kernel resample.hip
find gpurun_out/pmc_$name -name "*.db" -delete
cp gpurun_out/summary/resample_*_pmc.json profiles/ 2>/dev/null; for f in profiles/resample_*_pmc.json; do mv "$f" "profiles/r04_$(basename $f)"; done
( time timeout 900 python bench.py ) > gpurun_out/r4/bench8.log 2> gpurun_out/r4/bench8.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench8.log > gpurun_out/r4/bench8.json; tail -5 gpurun_out/r4/bench8.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench8.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', json.dumps(j['roofline'])[:300])
print('config', json.dumps(j['config'])[:2500])
for k, v in j['resample'].items():
    if isinstance(v, dict): print(k, v['ms_per_pass'], v['roofline']['frac'], v['within_2e-5'], v['roofline']['traffic'])
for k in ('e2e_8h_batch', 'e2e_16x1h', 'ahc_batch'):
    print(k, json.dumps(j.get(k))[:1600])
PY
