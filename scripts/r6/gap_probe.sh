#!/bin/bash
# round 6: how long the device idles between two replays of the round graph (host reads the state back, then launches the next replay)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_gap" -o gap -- python "$GRAFT_REPO_ROOT/scripts/ahc_probe.py" 43200 --kinds mix --check 0 ) > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3
import numpy as np
db = glob.glob('gpurun_out/prof_gap/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = list(cur.execute(f"select d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id where s.kernel_name like '%ahc_round_t%' order by d.start"))
a = np.array(rows, dtype=np.int64)
gap = (a[1:, 0] - a[:-1, 1]) / 1e3
dur = (a[:, 1] - a[:, 0]) / 1e3
big = gap[gap > 10]
print(f"rounds {len(a)}  duration us: mean {dur.mean():.2f} median {np.median(dur):.2f}  gap us: median {np.median(gap):.2f} mean {gap.mean():.2f}")
print(f"gaps > 10 us: {len(big)}  their mean {big.mean():.1f} us  sum {big.sum() / 1e3:.2f} ms  (of {(a[-1, 1] - a[0, 0]) / 1e6:.1f} ms)")
print("histogram of the big gaps (us):", np.percentile(big, [0, 25, 50, 75, 100]).round(1).tolist())
PY
rm -rf gpurun_out/prof_gap
