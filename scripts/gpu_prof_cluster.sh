#!/bin/bash
# rocprofv3 kernel trace of the clustering stage (8 h session, 4 calls)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/summary
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_cl" -o cl -- python "$GRAFT_REPO_ROOT/scripts/cluster_stage_probe.py" ) > gpurun_out/prof_cl.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py gpurun_out/prof_cl/cl_results.db --top 16 | tee gpurun_out/summary/cluster_stage_kernel_stats.txt
rm -rf gpurun_out/prof_cl
