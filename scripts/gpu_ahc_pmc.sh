#!/bin/bash
# PMC pass for the AHC start-up contraction (ahc_gram_mfma): MFMA busy cycles vs kernel cycles (--pmc only, no tracing combined)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc_ahc
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > gpurun_out/pmc_ahc/mfma_counters.txt
CMD="python $GRAFT_REPO_ROOT/scripts/ahc_probe.py 50000 --kinds iid --modes 0 --check 0"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_ahc/m1" -o m1 -- $CMD ) > gpurun_out/pmc_ahc/m1.log 2>&1; echo "m1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_ahc/m2" -o m2 -- $CMD ) > gpurun_out/pmc_ahc/m2.log 2>&1; echo "m2 rc=$?"
python scripts/pmc_summary.py ahc_gram_mfma $(find gpurun_out/pmc_ahc -name "*.db") > gpurun_out/pmc_ahc/gram_pmc.json
cat gpurun_out/pmc_ahc/mfma_counters.txt | tr '\n' ' '; echo
python - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_ahc/gram_pmc.json'))['counters']
for k, v in d.items(): print(k, v['per_dispatch'])
if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
    busy = d['SQ_VALU_MFMA_BUSY_CYCLES']['per_dispatch']; gui = d['GRBM_GUI_ACTIVE']['per_dispatch']
    print('GRBM_GUI_ACTIVE is summed over 8 XCDs; kernel cycles =', gui / 8)
    print('MFMA busy fraction (busy / (kernel cycles * 256 CUs * 4 SIMDs)) =', busy / (gui / 8 * 256 * 4))
PY
find gpurun_out/pmc_ahc -name "*.db" -delete
