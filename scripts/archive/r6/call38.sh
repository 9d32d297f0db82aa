#!/bin/bash
# round 6, call 38: slots per thread of the uniform-batch round (FA_AHC_UNI_CPT = 1 / 2 / 4) after the request diet, K = 8 / 12 resident
cd "$GRAFT_REPO_ROOT" || exit 1
{
for c in 1 2 4; do echo "## FA_AHC_UNI_CPT=$c"; FA_AHC_UNI_CPT=$c python scripts/r6/batch_groups_probe.py 8,12 0 --dev 2>&1 | grep -v amdgpu.ids; done
} | tee gpurun_out/r06_uni_cpt_after_diet.txt
