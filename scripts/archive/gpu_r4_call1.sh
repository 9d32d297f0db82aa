#!/bin/bash
# round 4, call 1: the whole GPU suite after the fa_ahc_cut / thread-start hardening, the bench line (incl. the self-launched 2-rank rehearsal
# inside the suite), the kernel trace of the headline step and the PMC passes of ahc_round_t for the new ahc.hip bytes.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r4/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r4/pytest_gpu.log | cut -c1-600
( time timeout 900 python bench.py ) > gpurun_out/r4/bench.log 2> gpurun_out/r4/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench.log > gpurun_out/r4/bench.json; cut -c1-2500 gpurun_out/r4/bench.json; tail -5 gpurun_out/r4/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam ) > gpurun_out/r4/rocprof_e2e.log 2>&1; echo "rocprof e2e rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 12 | tee gpurun_out/summary/e2e_kernel_stats.txt
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam
rm -rf gpurun_out/prof_e2e
