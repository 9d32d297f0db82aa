#!/bin/bash
# round 6, third final pass (the last tree: + the centroid sums without selects, the poisoned build's start-up): GPU suite on both builds, smoke(), the PMC pass of the
# round kernel for the present sources, the bench in the driver's command form, the kernel trace of the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6c gpurun_out/summary
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6c/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r6c/pytest.log | cut -c1-300
( time FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_poison.so python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6c/pytest_poison.log 2>&1; echo "pytest poison rc=$?"; tail -n 4 gpurun_out/r6c/pytest_poison.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc_round_body.h ahc_ws.h ahc_rounds.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ctc --skip-cpu --skip-ahc --skip-e2e --skip-beam --skip-resample 2>&1 | tail -n 2 | cut -c1-600
cp gpurun_out/summary/ahc_round_pmc.json profiles/r06_ahc_round_pmc.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6c/bench.out 2> gpurun_out/r6c/bench.err; echo "bench rc=$?"; tail -n 3 gpurun_out/r6c/bench.err
cp bench_legs.json gpurun_out/r6c/bench_legs.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r6c/rocprof_e2e.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py $(find gpurun_out/prof_e2e -name "*.db" | head -n 1) --top 14 | tee gpurun_out/summary/r06_e2e_kernel_stats.txt | cut -c1-200
rm -rf gpurun_out/prof_e2e
tail -n 1 gpurun_out/r6c/bench.out | cut -c1-2500
