#!/bin/bash
# round 5, call 17: the reference-order run through the matrix filter — both forms against the reference build, then the 8 h session probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
( time timeout 400 python -m pytest tests/test_gpu_ahc_adversarial.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "both_reference_order_forms or reference_order_mode or batch_with_tied" ) > gpurun_out/r5/pytest17.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r5/pytest17.log | cut -c1-600
FA_AHC_DEBUG=1 timeout 300 python scripts/r5/rom_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/rom_probe.log | cut -c1-400
