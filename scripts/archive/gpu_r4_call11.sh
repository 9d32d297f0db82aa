#!/bin/bash
# round 4, call 11: host-side changes of ahc.hip (a capped context runs one batch at a time; a failing set-up fails the batch): linkage / workspace / pool /
# pipeline tests, PMC passes for the present bytes of ahc.hip
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ahc.py tests/test_gpu_workspace.py tests/test_gpu_pool.py tests/test_gpu_pipeline.py tests/test_gpu_e2e_digest.py -m gpu -q --timeout=600 -p no:cacheprovider ) > gpurun_out/r4/pytest_call11.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r4/pytest_call11.log | cut -c1-700
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample
name=r4k
mkdir -p gpurun_out/pmc_$name
runp() { n=$1; shift; ( cd /tmp && FA_PROBE=uni FA_PROBE_K=4 timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python scripts/pmc_summary.py ahc_round_uni $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/ahc_round_uni_k4_pmc.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
p = 'gpurun_out/summary/ahc_round_uni_k4_pmc.json'
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256(('ahc.hip',))
j['kernel_sources'] = ['ahc.hip']
json.dump(j, open(p, 'w'), indent=1)
print('uni k4', {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256')})
PY
find gpurun_out/pmc_$name -name "*.db" -delete
