#!/bin/bash
# round 5, call 15 (re-run after every change of resample.hip): resampler PMC passes (bench.py's traffic, SHA-gated) on the final resample.hip, then the default bench with kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5 gpurun_out/summary
name=r5k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=resample timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
runp sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH
runp sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU
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
summ resample_44100 poly_rows_wide16_kernelILi16 "resample.hip resample_geom.h"
summ resample_22050 poly_rows_wide16_kernelILi10 "resample.hip resample_geom.h"
summ resample_8000 poly_interp_kernel "resample.hip resample_geom.h"
summ resample_48000 poly_decim_tile_kernel "resample.hip resample_geom.h"
find gpurun_out/pmc_$name -name "*.db" -delete
( time timeout 1200 python bench.py ) > gpurun_out/r5/bench15.log 2> gpurun_out/r5/bench15.err; echo "bench rc=$?"
tail -1 gpurun_out/r5/bench15.log > gpurun_out/r5/bench15.json; tail -4 gpurun_out/r5/bench15.err
