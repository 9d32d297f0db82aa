#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into the text table kept under profiles/.
usage: rocprof_summary.py <results.db> [--top N]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 15
    cur = sqlite3.connect(db).cursor()
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                  max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), max(d.private_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} {'grid':>9} {'wg':>5} {'lds':>7} {'scratch':>7}")
    for r in rows[:top]:
        name = r[0].replace(".kd", "")
        if len(name) > 68:
            name = name[:65] + "..."
        print(f"{name:<70} {r[1]:>7} {r[2] / 1e6:>10.3f} {r[3] / 1e3:>10.2f} {r[4] / 1e3:>10.2f} {r[5] / 1e3:>10.2f} {100 * r[2] / total:>6.1f} {r[6]:>9} {r[7]:>5} {r[8]:>7} {r[9]:>7}")


if __name__ == "__main__":
    main()
