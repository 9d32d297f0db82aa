"""Runs the REFERENCE's own centroid linkage (oracle/_ref = FastClusterWrapper.cpp compiled from /root/reference by
oracle/Makefile) on the full-size inputs of BASELINE configs[2] and commits digests of what it returns.

    python tests/golden/make_ahc_full_digest.py --dist iid --n 50000      (≈17-25 CPU-minutes, one core)
    python tests/golden/make_ahc_full_digest.py --tied dup30              (43 200 x 256 with exact ties; ≈10 CPU-minutes)

Writes tests/golden/ahc_full_<dist>_<n>.json  (SHA-256 of the dendrogram bytes, of the merge pairs / heights / sizes, of
the label vectors after AHCClustering's cut at thr 0.6 / 1.0 / 1.05 / 1.2, cluster counts, wall-clock of the reference)
and tests/golden/ahc_full_<dist>_<n>_pairs.npz (the merge pairs as int32, so that a device mismatch can be located).
The GPU parity tests and bench.py regenerate the same input bytes (ahc_full_inputs.py) and compare digests; nothing at
GPU-test / bench time needs /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import oracle  # noqa: E402
from ahc_full_inputs import THRESHOLDS, TIED_KINDS, ahc_input, ahc_tied_input, dendrogram_digest, sha256  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", choices=["iid", "mix"])
    ap.add_argument("--n", type=int)
    ap.add_argument("--tied", choices=list(TIED_KINDS), help="an input WITH exact ties at the size of configs[4] (ahc_full_inputs.ahc_tied_input); "
                                                             "writes ahc_tied_<kind>_<n>.json / _pairs.npz")
    ap.add_argument("--hours", type=float, default=8.0, help="--tied: length of the session (5 400 rows per hour)")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--stem", default=None, help="file stem under tests/golden (default ahc_full_<dist>_<n>); the matrix-free case: ahc_mf_iid_200000x4 with --d 4")
    a = ap.parse_args()
    oracle.build()
    assert oracle.ref_available(), "oracle/_ref is not built (needs /root/reference)"
    if a.tied:
        x = ahc_tied_input(a.tied, a.hours)
        a.dist, a.n, a.d = "tied_" + a.tied, len(x), x.shape[1]
        a.stem = a.stem or f"ahc_tied_{a.tied}_{a.n}"
    else:
        assert a.dist and a.n, "--dist and --n, or --tied"
        x = ahc_input(a.dist, a.n, a.d)
    t0 = time.perf_counter()
    st, z = oracle.linkage_ref(x)
    wall = time.perf_counter() - t0
    assert st == 0, st
    out = {"dist": a.dist, "n": a.n, "d": a.d, "seed": 0, "input_sha256": sha256(x), "numpy": np.__version__,
           "reference": "fastcluster_compute_centroid_linkage, FastClusterWrapper.cpp:196-244 built -O2 (oracle/Makefile)",
           "reference_seconds_1_core": wall, **dendrogram_digest(z), "cuts": {}}
    for thr in THRESHOLDS:
        lab = oracle.ahc_cut(z, a.n, thr)
        out["cuts"][repr(thr)] = {"labels_sha256": sha256(lab.astype(np.int32)), "clusters": int(lab.max()) + 1}
    stem = os.path.join(HERE, a.stem or f"ahc_full_{a.dist}_{a.n}")
    with open(stem + ".json", "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(stem + "_pairs.npz", pairs=z[:, :2].astype(np.int32))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
