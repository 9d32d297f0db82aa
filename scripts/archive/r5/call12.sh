#!/bin/bash
# round 5, call 12: the whole GPU suite, smoke(), PMC passes for the present bytes of ahc.hip / ctc.hip / tdt.hip / resample.hip, the bench line, the kernel trace
# of the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 -p no:cacheprovider ) > gpurun_out/r5/pytest12.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r5/pytest12.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample
bash scripts/gpu_pmc_kernel.sh ctc ctc_greedy "ctc.hip" python $GRAFT_REPO_ROOT/scripts/ctc_probe.py
name=r5k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=resample,tdt,uni FA_PROBE_K=4 timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
summ() {  # <output stem> <kernel pattern> "<source files>"
  python scripts/pmc_summary.py "$2" $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/$1_pmc.json
  python - "$1" "$3" <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
p = f'gpurun_out/summary/{sys.argv[1]}_pmc.json'
srcs = tuple(sys.argv[2].split())
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256(srcs)
j['kernel_sources'] = list(srcs)
c = j['counters']
if 'SQ_WAVES' in c and c['SQ_WAVES']['per_dispatch'] > 0:
    w = c['SQ_WAVES']['per_dispatch']
    j['instructions_per_wavefront'] = {k[9:].lower(): c[k]['per_dispatch'] / w for k in c if k.startswith('SQ_INSTS_')}
json.dump(j, open(p, 'w'), indent=1)
print(sys.argv[1], {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256')})
PY
}
summ resample_44100 poly_rows_kernelILi16 "resample.hip resample_geom.h"
summ resample_22050 poly_rows_kernelILi10 "resample.hip resample_geom.h"
summ resample_8000 poly_interp_kernel "resample.hip resample_geom.h"
summ resample_48000 poly_decim_kernel "resample.hip resample_geom.h"
summ tdt tdt_logits_fits_kernel tdt.hip
summ ahc_round_uni_c2_k4 ahc_round_uni_c2 ahc.hip
find gpurun_out/pmc_$name -name "*.db" -delete
( time timeout 1200 python bench.py ) > gpurun_out/r5/bench12.log 2> gpurun_out/r5/bench12.err; echo "bench rc=$?"
tail -1 gpurun_out/r5/bench12.log > gpurun_out/r5/bench12.json; tail -4 gpurun_out/r5/bench12.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r5/rocprof_e2e12.log 2>&1; echo "rocprof e2e rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 14 | tee gpurun_out/summary/r05_e2e_kernel_stats.txt
rm -rf gpurun_out/prof_e2e
python - <<'PY'
import json
j = json.load(open('gpurun_out/r5/bench12.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roof', json.dumps(j['roofline'])[:500])
print('config', json.dumps(j['config'])[1400:4200])
PY
