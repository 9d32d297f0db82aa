#!/bin/bash
# round 5, call 13: instruction mix of the wide row kernels at 44.1 kHz (PMC): FA_DBG=0 (whole kernel) and 4 (arithmetic only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/summary
for cfg in "16 8 0" "16 8 4" "32 8 4"; do
  set -- $cfg
  tag=w$1_$2_dbg$3
  export FA_RESAMPLE_WIDE_ROWS=$1 FA_RESAMPLE_WIDE_WAVES=$2 FA_DBG=$3
  mkdir -p gpurun_out/pmc_$tag
  run() { n=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r5/resample_one.py 44100 160 441 ) > gpurun_out/pmc_$tag/$n.log 2>&1; echo "$tag/$n rc=$?"; }
  run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH
  run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU
  python scripts/pmc_summary.py poly_rows_wide $(find gpurun_out/pmc_$tag -name "*.db") > gpurun_out/summary/r05_wide_${tag}_pmc.json
  python - gpurun_out/summary/r05_wide_${tag}_pmc.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[1], {k: round(v['per_dispatch']) for k, v in j['counters'].items()})
PY
  find gpurun_out/pmc_$tag -name "*.db" -delete
done
