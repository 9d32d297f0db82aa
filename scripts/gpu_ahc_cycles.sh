#!/bin/bash
# cycle-stamp build of the AHC round kernel (diagnostic only; the .so of the snapshot on the box is rebuilt with -DFA_AHC_PROFILE)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( cd fluidaudio_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DFA_AHC_PROFILE=1 -c ahc.hip -o ahc.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfluidaudio_hip.so ctx.o pool.o mel.o ctc.o beam.o tdt.o ahc.o vbx.o post.o kmeans.o resample.o formats.o offline.o ) && timeout 300 python scripts/ahc_probe.py 50000 --kinds iid --modes 0 --check 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ahc_cycles.log
