#!/bin/bash
# round 6, call 12: fp16 TDT rows through pair requests (nine dwords per lane instead of seventeen halves): parity + the leg
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_tdt.py -q -p no:cacheprovider 2>&1 | tail -n 3
python scripts/tdt_leg_probe.py 1024:float16,4096:float16,8192:float16,1024:float32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_tdt_pairs_probe.txt
