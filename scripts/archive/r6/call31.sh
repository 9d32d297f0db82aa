#!/bin/bash
# round 6, call 31: runtime knobs against the kernel boundary of the round chain: where kernel arguments live (HIP_FORCE_DEV_KERNARG), interrupt vs polled completion
cd "$GRAFT_REPO_ROOT" || exit 1
{
for env in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0"; do
  echo "## env: $env"
  for rep in 1 2; do env $env python scripts/ahc_probe.py 43200 --kinds mix --check 0 2>&1 | grep -v amdgpu.ids | cut -c1-60,230-420; done
done
} | tee gpurun_out/r06_runtime_knobs.txt
