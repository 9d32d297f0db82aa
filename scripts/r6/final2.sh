#!/bin/bash
# round 6, second final pass (after the request diet of the round kernel, fa_offline_cluster_batch_dev, fp16 TDT pairs): the GPU suite (release + poisoned workspace),
# smoke(), PMC passes for the sources that changed (round kernel single + uniform batch, TDT), the bench in the driver's command form, the kernel trace of the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6b gpurun_out/summary
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6b/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r6b/pytest.log | cut -c1-300
( time FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_poison.so python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6b/pytest_poison.log 2>&1; echo "pytest poison rc=$?"; tail -n 4 gpurun_out/r6b/pytest_poison.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc_round_body.h ahc_ws.h ahc_rounds.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ctc --skip-cpu --skip-ahc --skip-e2e --skip-beam --skip-resample 2>&1 | tail -n 2 | cut -c1-600
FA_PROBE=uni FA_PROBE_K=4 bash scripts/gpu_pmc_kernel.sh ahc_round_uni_c2_k4 ahc_round_uni_c2 "ahc_round_body.h ahc_ws.h ahc_batch.hip" python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py 2>&1 | tail -n 1 | cut -c1-600
FA_PROBE=tdt bash scripts/gpu_pmc_kernel.sh tdt tdt_logits_fits_kernel "tdt.hip" python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py 2>&1 | tail -n 1 | cut -c1-400
cp gpurun_out/summary/ahc_round_pmc.json profiles/r06_ahc_round_pmc.json; cp gpurun_out/summary/tdt_pmc.json profiles/r06_tdt_pmc.json; cp gpurun_out/summary/ahc_round_uni_c2_k4_pmc.json profiles/r06_ahc_round_uni_c2_k4_pmc.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6b/bench.out 2> gpurun_out/r6b/bench.err; echo "bench rc=$?"; tail -n 3 gpurun_out/r6b/bench.err
cp bench_legs.json gpurun_out/r6b/bench_legs.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_e2e" -o e2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --skip-mel --skip-ahc --skip-ctc --skip-cpu --skip-e2e --skip-beam --skip-resample ) > gpurun_out/r6b/rocprof_e2e.log 2>&1; echo "rocprof rc=$?"
python scripts/rocprof_summary.py $(find gpurun_out/prof_e2e -name "*.db" | head -n 1) --top 14 | tee gpurun_out/summary/r06_e2e_kernel_stats.txt | cut -c1-200
rm -rf gpurun_out/prof_e2e
tail -n 1 gpurun_out/r6b/bench.out | cut -c1-2500
