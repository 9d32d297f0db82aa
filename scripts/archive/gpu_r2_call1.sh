#!/bin/bash
# round 2, call 1: microbenchmarks + the GPU test-suite (with the new float64 / digest gates) + a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
timeout 120 scripts/ubench/valu_lds > gpurun_out/r2/ubench_valu_lds.txt 2>&1; echo "ubench rc=$?"
cat gpurun_out/r2/ubench_valu_lds.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider -x --deselect tests/test_gpu_ahc.py::test_full_size_digests_are_committed -s ) > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "bit-exact vs the reference|max rel err vs float64|passed|failed|error" gpurun_out/r2/pytest_gpu.log | tail -20
( time timeout 600 python bench.py --skip-e2e ) > gpurun_out/r2/bench.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/r2/bench.log | cut -c1-2500
