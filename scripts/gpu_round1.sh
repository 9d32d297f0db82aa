#!/bin/bash
# one gpurun call: smoke, gpu tests, bench, rocprof of the mel bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/smoke.log
( time timeout 1200 python -m pytest tests -m gpu -q --timeout=240 -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_gpu.log
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -5 gpurun_out/bench.log
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_mel" -o mel -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --skip-ahc --skip-ctc --skip-cpu ) > gpurun_out/rocprof_mel.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_mel -name "*stats*" | head
