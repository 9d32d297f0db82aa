#!/bin/bash
# round 3: the smaller kernels touched this round (resampler, centroids, Hungarian slabs) — tests + probes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
( timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_post.py tests/test_gpu_pipeline.py tests/test_gpu_e2e_digest.py -m gpu -q --timeout=600 -p no:cacheprovider ) 2>&1 | tail -15 | cut -c1-300
timeout 300 python scripts/resample_probe.py 2>/dev/null | tee gpurun_out/r3/resample_probe.json | cut -c1-1800
timeout 300 python scripts/cluster_stage_probe.py 2>&1 | tail -5 | cut -c1-1500
