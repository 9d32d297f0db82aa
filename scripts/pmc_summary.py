#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite) for one kernel into JSON.
usage: pmc_summary.py <kernel-substring> <db> [<db> ...]
Counters are summed over all hardware instances of a dispatch and averaged over dispatches.  FETCH_SIZE / WRITE_SIZE are
reported by rocprofv3 in KiB; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 tallies
128-byte requests at 64 bytes, so fetch_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is uncalibrated and taken as is."""
import json
import sqlite3
import sys


def main():
    pat, out = sys.argv[1], {}
    for db in sys.argv[2:]:
        cur = sqlite3.connect(db).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if "pmc_event" in t]
        info = [t for t in tabs if "info_pmc" in t]
        disp = [t for t in tabs if "kernel_dispatch" in t]
        sym = [t for t in tabs if "info_kernel_symbol" in t]
        if not (pmc and info and disp and sym):
            continue
        q = (f"select i.name, count(distinct d.id), sum(e.value) from {pmc[0]} e join {info[0]} i on e.pmc_id=i.id "
             f"join {disp[0]} d on e.event_id=d.event_id join {sym[0]} s on d.kernel_id=s.id "
             f"where s.kernel_name like '%{pat}%' group by i.name")
        for name, nd, total in cur.execute(q):
            out[name] = {"dispatches": nd, "per_dispatch": total / max(1, nd)}
    res = {"kernel": pat, "counters": out}
    if "FETCH_SIZE" in out:
        res["fetch_bytes_per_launch"] = 2.0 * 1024.0 * out["FETCH_SIZE"]["per_dispatch"]
    if "WRITE_SIZE" in out:
        res["write_bytes_per_launch"] = 1024.0 * out["WRITE_SIZE"]["per_dispatch"]
    if "fetch_bytes_per_launch" in res and "write_bytes_per_launch" in res:
        res["hbm_traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    if "SQ_WAVE_CYCLES" in out:
        wc = out["SQ_WAVE_CYCLES"]["per_dispatch"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in out:
                res[k + "_frac_of_wave_cycles"] = out[k]["per_dispatch"] / wc
    if "SQ_LDS_IDX_ACTIVE" in out and "SQ_LDS_BANK_CONFLICT" in out and out["SQ_LDS_IDX_ACTIVE"]["per_dispatch"] > 0:
        res["lds_bank_conflict_frac"] = out["SQ_LDS_BANK_CONFLICT"]["per_dispatch"] / out["SQ_LDS_IDX_ACTIVE"]["per_dispatch"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
