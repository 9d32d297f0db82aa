#!/bin/bash
# round 5, call 25: PMC passes for the two kernels of the matrix-filtered reference-order run (HBM traffic, instructions per wavefront), one 8 h session
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
name=rom
mkdir -p gpurun_out/pmc_$name gpurun_out/summary
runp() { n=$1; shift; ( cd /tmp && ROM_PROBE_ONLY=1 timeout 300 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r5/rom_probe.py ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
runp tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
runp tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
runp sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
for k in rom_scan rom_select; do
  python scripts/pmc_summary.py "$k" $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/r05_${k}_pmc.json
  python - "$k" <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
k = sys.argv[1]
p = f'gpurun_out/summary/r05_{k}_pmc.json'
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256(("ahc.hip", "ahc_reforder.h"))
j['kernel_sources'] = ["ahc.hip", "ahc_reforder.h"]
c = j['counters']
if 'SQ_WAVES' in c and c['SQ_WAVES']['per_dispatch'] > 0:
    w = c['SQ_WAVES']['per_dispatch']
    j['waves_per_launch'] = w
    j['instructions_per_wavefront'] = {q[9:].lower(): c[q]['per_dispatch'] / w for q in c if q.startswith('SQ_INSTS_')}
json.dump(j, open(p, 'w'), indent=1)
print(k, {q: v for q, v in j.items() if q not in ('counters', 'kernel_sources_sha256')})
PY
done
find gpurun_out/pmc_$name -name "*.db" -delete
