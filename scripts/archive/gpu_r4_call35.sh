#!/bin/bash
# round 4, call 35: the driver's own entry point on the final tree: smoke() (build() runs on the CPU side; the prebuilt library travels with the snapshot)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" ) > gpurun_out/r4/smoke35.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r4/smoke35.log | cut -c1-300
