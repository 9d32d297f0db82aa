#!/bin/bash
# round 6, call 3: the GPU suite on the split linkage units (release + poisoned workspace), then the PMC pass of the headline's round kernel for the present sources
python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -n 4
FLUIDAUDIO_HIP_LIBRARY=$PWD/fluidaudio_amd/csrc/libfluidaudio_hip_poison.so python -m pytest tests -q -m gpu -p no:cacheprovider -k "ahc or e2e or pipeline or workspace or degrade or pool" 2>&1 | tail -n 4
bash scripts/gpu_pmc_kernel.sh ahc_round ahc_round_t "ahc_round_body.h ahc_ws.h ahc_rounds.hip" python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --skip-mel --skip-ctc --skip-cpu --skip-ahc --skip-e2e --skip-beam --skip-resample 2>&1 | tail -n 8
