#!/usr/bin/env python3
"""How many merges of the reference's merge order could ONE launch of the round kernel take?  (CPU study for DESIGN.md §6; uses the
reference linkage build under oracle/_ref as the source of the merge order — test infrastructure, not the product.)

The merge chain of one recording is one kernel launch per merge (43 200 launches of 5.26 us for the 8 h session: 91 % of the headline).
Greedy centroid linkage merges the globally closest pair, so merge i + 1 depends on merge i only through the NEW node's row: if the pair
of merge i + 1 does not contain the node merge i created, it is the closest pair among the nodes that existed before merge i and are
disjoint from its pair — known BEFORE merge i runs.  A launch could therefore take the k closest mutually disjoint pairs and merge them
all, provided a check AFTER the fact holds: no entry of a row created in the launch lies below the height of a later merge of the same
launch (otherwise the reference would have merged that entry first, and the launch must be replayed from its first invalid merge).

This script measures the ceiling of that idea on the reference's own merge order: the merge sequence is cut greedily into batches whose
merges only use nodes that existed when the batch began (every such batch passes the check by construction), with the batch size
capped at k.  launches(k) / merges is the fraction of kernel launches left.

usage: merge_independence.py [--hours 8 | --n 50000 --dist iid|mixture] [--sigma 0.03]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def batches(z, n, cap):
    """number of launches when a launch takes up to `cap` consecutive merges none of which uses a node created in the same launch"""
    launches, start, i, m = 0, 0, 0, len(z)
    a = z[:, 0].astype(np.int64)
    b = z[:, 1].astype(np.int64)
    while i < m:
        launches += 1
        first_new = n + i                      # nodes >= first_new are created in this launch
        j = i
        while j < m and j - i < cap and a[j] < first_new and b[j] < first_new:
            j += 1
        if j == i:                             # cannot happen: merge i only uses nodes < n + i
            j = i + 1
        i = j
    return launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=float, default=None)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--dist", default="mixture")
    ap.add_argument("--sigma", type=float, default=0.03)
    ap.add_argument("--out")
    a = ap.parse_args()
    import oracle
    if a.hours is not None:
        from e2e_inputs import e2e_session
        s = e2e_session(a.hours, 12, seed=5, sigma=a.sigma)
        x = oracle.ahc_normalize(s["emb"].astype(np.float64))
        name = f"e2e session {a.hours:g} h, sigma {a.sigma:g} (N = {len(x)})"
    elif a.dist == "iid":
        x = oracle.ahc_normalize(np.random.default_rng(0).standard_normal((a.n, 256)))
        name = f"iid N(0, 1) rows (N = {a.n})"
    else:
        from conftest import speaker_mixture
        x = speaker_mixture(a.n, 256, 64, 0.02, seed=0)
        name = f"64-speaker mixture, sigma 0.02 (N = {a.n})"
    n = len(x)
    t0 = time.time()
    st, z = oracle.linkage_ref(x)
    assert st == 0
    dt = time.time() - t0
    out = {"input": name, "merges": n - 1, "reference_linkage_s": round(dt, 1), "launches": {}}
    for cap in (1, 2, 3, 4, 8, 16, 64, 10 ** 9):
        L = batches(z, n, cap)
        out["launches"]["unbounded" if cap > 10 ** 6 else str(cap)] = {"launches": L, "fraction_of_merges": round(L / (n - 1), 4)}
    # where in the run the dependent merges sit: fraction of merges that use the node created by the merge just before them, by decile
    a_ = z[:, 0].astype(np.int64)
    b_ = z[:, 1].astype(np.int64)
    prev_new = n + np.arange(n - 1) - 1
    dep = (a_ == prev_new) | (b_ == prev_new)
    out["uses_previous_node_by_decile"] = [round(float(dep[i * (n - 1) // 10:(i + 1) * (n - 1) // 10].mean()), 4) for i in range(10)]
    out["uses_previous_node"] = round(float(dep.mean()), 4)
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
