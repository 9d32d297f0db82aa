#!/bin/bash
# round 6, last pass on the last tree (the Gram start-up launched over the triangular tile set): GPU suite on both builds, smoke(), the bench in the driver's command form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6d
( time python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6d/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r6d/pytest.log | cut -c1-300
( time FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_poison.so python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r6d/pytest_poison.log 2>&1; echo "pytest poison rc=$?"; tail -n 4 gpurun_out/r6d/pytest_poison.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6d/bench.out 2> gpurun_out/r6d/bench.err; echo "bench rc=$?"; tail -n 3 gpurun_out/r6d/bench.err
cp bench_legs.json gpurun_out/r6d/bench_legs.json
tail -n 1 gpurun_out/r6d/bench.out | cut -c1-2500
