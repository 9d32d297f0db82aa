#!/bin/bash
# round 4, call 7: the whole GPU suite on the final tree, per-kernel PMC passes of the resamplers / the TDT walk / the uniform round (with source
# SHAs, so that bench.py can attach the traffic), the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider ) > gpurun_out/r4/pytest_call7.log 2>&1; echo "pytest rc=$?"
( time timeout 900 python scripts/uni_probe.py ) > gpurun_out/r4/uni_probe.log 2>&1; echo "uni probe rc=$?"; grep "^{" gpurun_out/r4/uni_probe.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['n'], r['K'], r['env'], 'us/round %.2f' % r['us_per_round'], 'wall %.3f' % r['wall_s'], 'h/s %.1f' % r.get('audio_hours_per_s_linkage_only', 0), r['equal_single'])
"; tail -3 gpurun_out/r4/uni_probe.log | cut -c1-300
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample
tail -12 gpurun_out/r4/pytest_call7.log | cut -c1-700
name=r4k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=resample,tdt,uni FA_PROBE_K=4 timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
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
print(sys.argv[1], {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256')}, {k: round(v['per_dispatch']) for k, v in j['counters'].items() if k.startswith('SQ_INSTS') or k == 'SQ_WAVES'})
PY
}
summ resample_44100 poly_rows_kernelILi16 resample.hip
summ resample_22050 poly_rows_kernelILi8 resample.hip
summ resample_8000 poly_interp_kernel resample.hip
summ resample_48000 poly_decim_kernel resample.hip
summ tdt tdt_logits_kernel tdt.hip
summ ahc_round_uni_k4 ahc_round_uni ahc.hip
find gpurun_out/pmc_$name -name "*.db" -delete
( time timeout 900 python bench.py ) > gpurun_out/r4/bench7.log 2> gpurun_out/r4/bench7.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench7.log > gpurun_out/r4/bench7.json; tail -5 gpurun_out/r4/bench7.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r4/bench7.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', json.dumps(j['roofline'])[:400])
print('config', json.dumps(j['config'])[:2500])
for k, v in j['resample'].items():
    if isinstance(v, dict): print(k, v['ms_per_pass'], v['roofline']['frac'], v['within_2e-5'], v['roofline']['traffic'])
for k in ('tdt', 'e2e_8h_batch', 'e2e_8h_hard', 'ctc_fp16'):
    print(k, json.dumps(j.get(k))[:900])
PY
