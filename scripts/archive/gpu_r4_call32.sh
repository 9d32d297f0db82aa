#!/bin/bash
# round 4, call 32: the whole GPU suite on the final sources (CTC with streaming loads, the rebuilt beam walk), PMC traffic of ctc_greedy_kernel for the
# present ctc.hip, the bench line, the kernel trace of the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4 gpurun_out/summary
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x ) > gpurun_out/r4/pytest_gpu32.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r4/pytest_gpu32.log | cut -c1-600
bash scripts/gpu_pmc_kernel.sh ctc ctc_greedy "ctc.hip" python $GRAFT_REPO_ROOT/scripts/ctc_probe.py
cp gpurun_out/summary/ctc_pmc.json profiles/r04_ctc_pmc.json
( time timeout 900 python bench.py ) > gpurun_out/r4/bench32.log 2> gpurun_out/r4/bench32.err; echo "bench rc=$?"
tail -1 gpurun_out/r4/bench32.log > gpurun_out/r4/bench32.json; cut -c1-1500 gpurun_out/r4/bench32.json; tail -3 gpurun_out/r4/bench32.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r4/rocprof_e2e32.log 2>&1; echo "rocprof e2e rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db --top 12 | tee gpurun_out/summary/e2e_kernel_stats32.txt
rm -rf gpurun_out/prof_e2e
