#!/bin/bash
# AHC round with fewer piggy-backed re-scans (-DFA_AHC_PIGGY=n): build on the box, time 50k, rebuild default
cd "$GRAFT_REPO_ROOT" || exit 1
for P in 1 3; do
( cd fluidaudio_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DFA_AHC_PIGGY=$P -c ahc.hip -o ahc.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfluidaudio_hip.so ctx.o pool.o mel.o ctc.o beam.o tdt.o ahc.o vbx.o post.o kmeans.o resample.o formats.o offline.o ) && echo "piggy $P" && timeout 300 python scripts/ahc_probe.py 50000 --kinds iid,mix --modes 0,1 --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
timeout 300 python scripts/ahc_probe.py 3000 --kinds iid,mix --modes 0,1 2>&1 | grep -v amdgpu.ids | cut -c1-250
