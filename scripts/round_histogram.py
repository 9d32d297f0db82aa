#!/usr/bin/env python3
"""Distribution of the launch durations and launch-to-launch periods of ahc_round_t in a rocprofv3 kernel trace, by position in the run
(the merge chain slows down where more partner entries are column copies).  usage: round_histogram.py <results.db>"""
import sqlite3
import sys

import numpy as np

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("""select d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                           where s.kernel_name like '%ahc_round_t%' order by d.start"""))
a = np.array(rows, dtype=np.int64)
dur = (a[:, 1] - a[:, 0]) / 1e3
per = np.diff(a[:, 0]) / 1e3
print(f"{len(dur)} launches; duration us: mean {dur.mean():.2f} p10 {np.percentile(dur, 10):.2f} p50 {np.percentile(dur, 50):.2f} p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} max {dur.max():.1f}")
ok = per < 100
print(f"period us (start to start, gaps > 100 us = graph replays dropped): mean {per[ok].mean():.2f} p50 {np.percentile(per[ok], 50):.2f} p90 {np.percentile(per[ok], 90):.2f}")
n = len(dur)
print("by tenth of the run: mean duration | mean period")
for k in range(10):
    lo, hi = k * n // 10, (k + 1) * n // 10
    p = per[lo:min(hi, len(per))]
    p = p[p < 100]
    print(f"  {k}: {dur[lo:hi].mean():.2f} | {p.mean():.2f}")
edges = [0, 2, 3, 3.5, 4, 4.5, 5, 6, 8, 12, 1e9]
h, _ = np.histogram(dur, edges)
print("duration histogram (us):", ", ".join(f"<{e:g}: {c}" for e, c in zip(edges[1:], h)))
