#!/bin/bash
# round 4, call 6: evidence — kernel traces (headline step; the round-4 kernels), PMC passes (ahc_round_t for the present ahc.hip bytes, the
# uniform round, the resamplers, the TDT walk), TDT / workspace tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_tdt.py tests/test_gpu_workspace.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call6.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r4/pytest_call6.log | cut -c1-700
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r4/rocprof_e2e.log 2>&1; echo "rocprof e2e rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 12 | tee gpurun_out/summary/e2e_kernel_stats.txt
( cd /tmp && FA_PROBE_K=8 timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r4k" -o r4k -- python "$GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py" ) > gpurun_out/r4/rocprof_r4k.log 2>&1; echo "rocprof r4 kernels rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_r4k/r4k_results.db --top 24 | tee gpurun_out/summary/r4_kernels_stats.txt
rm -rf gpurun_out/prof_e2e gpurun_out/prof_r4k
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample
# the round-4 kernels: ONE set of PMC passes over the probe, summarised per kernel
name=r4k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=resample,tdt,uni FA_PROBE_K=4 timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
for k in ahc_round_uni poly_rows_kernel poly_interp_kernel poly_decim_kernel tdt_logits_kernel; do
  python scripts/pmc_summary.py "$k" $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/${k}_pmc.json
  python - "$k" <<'PY'
import json, sys
j = json.load(open(f'gpurun_out/summary/{sys.argv[1]}_pmc.json'))
print(sys.argv[1], {k: v for k, v in j.items() if k != 'counters'}, {k: round(v['per_dispatch']) for k, v in j['counters'].items() if k.startswith('SQ_INSTS') or k == 'SQ_WAVES'})
PY
done
find gpurun_out/pmc_$name -name "*.db" -delete
