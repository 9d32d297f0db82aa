#!/bin/bash
# rocprofv3 kernel trace of one 43 200 x 256 linkage problem: start-up kernels + the distribution of the round launches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/summary
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ai" -o ai -- python "$GRAFT_REPO_ROOT/scripts/ahc_probe.py" 43200 --kinds e2e --modes 0 --check 0 ) > gpurun_out/prof_ai.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_ai/ai_results.db --top 8 | tee gpurun_out/summary/ahc_init_kernel_stats.txt
python scripts/round_histogram.py gpurun_out/prof_ai/ai_results.db | tee gpurun_out/summary/ahc_round_histogram.txt
rm -rf gpurun_out/prof_ai
