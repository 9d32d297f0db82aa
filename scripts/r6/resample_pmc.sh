#!/bin/bash
# HBM traffic + issue / LDS counters of the resampler's main kernel for every rate of the bench leg (separate --pmc passes, no tracing combined):
# gpurun_out/summary/r06_resample_<rate>_pmc.json, each with the SHA-256 of the kernel's sources (bench.py refuses stale numbers).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/summary
for rate in ${RATES:-48000 44100 22050 8000 96000 88200}; do
  d=gpurun_out/pmc_rs_$rate; mkdir -p $d
  run() { n=$1; shift; ( cd /tmp && FA_PROBE_RATE=$rate timeout 300 rocprofv3 --pmc "$@" -d "$GRAFT_REPO_ROOT/$d/$n" -o $n -- python $GRAFT_REPO_ROOT/scripts/r6/resample_probe.py ) > $d/$n.log 2>&1; echo "$rate/$n rc=$?"; }
  run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
  run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
  run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  run sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  python - "$rate" $(find $d -name "*.db") <<'PY'
import json, sqlite3, subprocess, sys
sys.path.insert(0, '.')
import bench
rate, dbs = sys.argv[1], sys.argv[2:]
# the main kernel of the rate: the poly_* kernel with the largest total time in the first database
cur = sqlite3.connect(dbs[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, sum(d.end - d.start), count(*) from {disp} d join {sym} s on d.kernel_id = s.id where s.kernel_name like '%poly_%' group by s.kernel_name order by 2 desc"))
main = rows[0][0]
pat = main.split('(')[0].replace('.kd', '')
pat = pat[:60]
out = subprocess.run([sys.executable, 'scripts/pmc_summary.py', pat] + dbs, capture_output=True, text=True).stdout
j = json.loads(out)
srcs = ("resample.hip", "resample_geom.h")
j['kernel_sources_sha256'] = bench.sources_sha256(srcs); j['kernel_sources'] = list(srcs)
j['kernels_of_the_pass'] = [{"name": r[0][:90], "total_us": r[1] / 1e3, "launches": r[2]} for r in rows]
c = j['counters']
if 'SQ_WAVES' in c and c['SQ_WAVES']['per_dispatch'] > 0:
    w = c['SQ_WAVES']['per_dispatch']
    j['instructions_per_wavefront'] = {k[9:].lower(): c[k]['per_dispatch'] / w for k in c if k.startswith('SQ_INSTS_')}
json.dump(j, open(f'gpurun_out/summary/r06_resample_{rate}_pmc.json', 'w'), indent=1)
print(rate, {k: v for k, v in j.items() if k not in ('counters', 'kernel_sources_sha256', 'kernels_of_the_pass')})
PY
  find $d -name "*.db" -delete
done
