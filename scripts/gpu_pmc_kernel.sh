#!/bin/bash
# HBM traffic of one kernel from rocprofv3 PMC passes (separate --pmc runs, no tracing combined), summarised to JSON with the SHA-256 of
# the kernel's sources so that bench.py refuses stale numbers.
#   usage: gpu_pmc_kernel.sh <name> <kernel-substring> "<source files under csrc>" <command...>
cd "$GRAFT_REPO_ROOT" || exit 1
name=$1; pat=$2; srcs=$3; shift 3
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$name gpurun_out/summary
run() { n=$1; shift; ( cd /tmp && timeout 600 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/$n" -o $n -- $CMD ) > gpurun_out/pmc_$name/$n.log 2>&1; echo "$name/$n rc=$?"; }
CMD="$*"
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python scripts/pmc_summary.py "$pat" $(find gpurun_out/pmc_$name -name "*.db") > gpurun_out/summary/${name}_pmc.json
python - "$name" "$srcs" <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
name, srcs = sys.argv[1], tuple(sys.argv[2].split())
p = f'gpurun_out/summary/{name}_pmc.json'
j = json.load(open(p))
j['kernel_sources_sha256'] = bench.sources_sha256(srcs)
j['kernel_sources'] = list(srcs)
json.dump(j, open(p, 'w'), indent=1)
print({k: v for k, v in j.items() if k != 'counters'})
PY
find gpurun_out/pmc_$name -name "*.db" -delete
