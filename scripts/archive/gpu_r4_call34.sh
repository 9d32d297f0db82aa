#!/bin/bash
# round 4, call 34: the whole GPU suite on the final sources, PMC traffic of tdt_logits_kernel for the present tdt.hip, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r4/pytest_gpu34.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4/pytest_gpu34.log | cut -c1-600
name=r4k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=tdt timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python scripts/pmc_summary.py tdt_logits_kernel $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/tdt_pmc.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
p = 'gpurun_out/summary/tdt_pmc.json'
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256(bench.TDT_SOURCES)
j['kernel_sources'] = list(bench.TDT_SOURCES)
json.dump(j, open(p, 'w'), indent=1)
print('tdt', {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256')}, {k: round(v['per_dispatch']) for k, v in j['counters'].items() if k.startswith('SQ_INSTS') or k == 'SQ_WAVES'})
PY
cp gpurun_out/summary/tdt_pmc.json profiles/r04_tdt_pmc.json
find gpurun_out/pmc_$name -name "*.db" -delete
( time timeout 900 python bench.py ) > gpurun_out/r4/bench34.log 2> gpurun_out/r4/bench34.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench34.log > gpurun_out/r4/bench34.json; tail -3 gpurun_out/r4/bench34.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench34.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'traffic', j['roofline']['traffic'])
print('tdt', json.dumps(j.get('tdt'))[:900])
PY
