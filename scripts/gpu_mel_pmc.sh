#!/bin/bash
# PMC passes for the mel kernel (separate runs per counter group; --pmc only, no tracing combined)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --only-mel --mel-steps 3 --mel-warmup 1 --clock-warm-s 0"
run() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name" -o $name -- $CMD ) > gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
python - <<'PY'
import sqlite3, glob, os
for db in sorted(glob.glob('gpurun_out/pmc/*/*.db')+glob.glob('gpurun_out/pmc/*/*/*.db')):
    cur = sqlite3.connect(db).cursor()
    tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc=[t for t in tabs if 'pmc_event' in t]; info=[t for t in tabs if 'info_pmc' in t]; disp=[t for t in tabs if 'kernel_dispatch' in t]; sym=[t for t in tabs if 'info_kernel_symbol' in t]
    if not pmc: print(db,'no pmc tables',tabs[:5]); continue
    q=f"""select s.kernel_name, i.name, count(*), sum(e.value) from {pmc[0]} e join {info[0]} i on e.pmc_id=i.id join {disp[0]} d on e.event_id=d.event_id join {sym[0]} s on d.kernel_id=s.id group by s.kernel_name, i.name"""
    try:
        for r in cur.execute(q):
            if 'mel_kernel' in r[0]: print(os.path.basename(db), r[1], 'dispatches', r[2], 'sum', r[3], 'per_dispatch', r[3]/r[2])
    except Exception as ex: print(db, 'query failed', ex, tabs)
PY
