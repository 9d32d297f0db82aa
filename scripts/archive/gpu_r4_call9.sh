#!/bin/bash
# round 4, call 9: final tree — whole GPU suite, PMC passes for the present bytes of ahc.hip (ahc_round_t; ahc_round_uni) and tdt.hip, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r4/pytest_call9.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r4/pytest_call9.log | cut -c1-700
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample
cp gpurun_out/summary/ahc_round_pmc.json profiles/r04_ahc_round_pmc.json
name=r4k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=tdt,uni FA_PROBE_K=4 timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
summ() {
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
print(sys.argv[1], {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256')}, {k: round(v['per_dispatch']) for k, v in j['counters'].items() if k.startswith('SQ_INSTS') or k == 'SQ_WAVES'})
PY
}
summ tdt tdt_logits_kernel tdt.hip
summ ahc_round_uni_k4 ahc_round_uni ahc.hip
cp gpurun_out/summary/tdt_pmc.json profiles/r04_tdt_pmc.json
find gpurun_out/pmc_$name -name "*.db" -delete
( time timeout 900 python bench.py ) > gpurun_out/r4/bench9.log 2> gpurun_out/r4/bench9.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench9.log > gpurun_out/r4/bench9.json; tail -5 gpurun_out/r4/bench9.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench9.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', json.dumps(j['roofline'])[:330])
print('config', json.dumps(j['config'])[1500:2600])
for k in ('tdt',):
    print(k, json.dumps(j.get(k))[:1000])
print({k: v['audio_hours_per_s'] for k, v in j['e2e_8h_batch'].items() if k.startswith('x')})
PY
