#!/bin/bash
# round 6, call 5: TDT walk with the first-index search in the vector unit: parity, leg timing, PMC (instructions per decision)
python -m pytest tests/test_gpu_tdt.py -q -p no:cacheprovider 2>&1 | tail -n 3
python scripts/tdt_leg_probe.py 2>&1 | grep '"B"' | cut -c1-300
export FA_PROBE=tdt
bash scripts/gpu_pmc_kernel.sh tdt tdt_logits_fits_kernel "tdt.hip" python $GRAFT_REPO_ROOT/scripts/r4_kernels_probe.py 2>&1 | tail -n 5 | cut -c1-1500
