#!/bin/bash
# round 5, call 7: instruction mix of poly_rows_kernel at 44.1 kHz (PMC)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_rs gpurun_out/summary
run() { n=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_rs/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r5/resample_one.py 44100 160 441 ) > gpurun_out/pmc_rs/$n.log 2>&1; echo "rs/$n rc=$?"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU
python scripts/pmc_summary.py poly_rows $(find gpurun_out/pmc_rs -name "*.db") > gpurun_out/summary/r05_rows_pmc.json
cat gpurun_out/summary/r05_rows_pmc.json | head -80
find gpurun_out/pmc_rs -name "*.db" -delete
