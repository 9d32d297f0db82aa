"""The tie route at the configured size against the REFERENCE BUILD (VERDICT r5 "missing" #2): 43 200 x 256 inputs with exact ties —
30 % duplicated rows, 5 % one identical row ("digital silence"), rows on a 1/64 grid — whose dendrograms the reference's own C++
(oracle/_ref, built from /root/reference) produced once on the CPU (tests/golden/make_ahc_full_digest.py --tied <kind>, ~8 CPU-minutes each);
SHA-256 digests + merge pairs are committed.  Which of the tied pairs the reference merges is its heap's order
(fastcluster_internal.hpp:1685-1799, heap ties :1705-1734, :1792-1797): the device must reproduce it row for row through
  AUTO                                 (the filter rounds halt at the first exact tie, the problem re-runs in reference order),
  REFERENCE_ORDER, matrix-filtered     (rom_scan / rom_select: Lance-Williams candidates, exact sums of the few, the key-carrying block heap),
  REFERENCE_ORDER, matrix-free         (FA_AHC_RO_NO_MATRIX: O(A d) exact sums per row, the restated heap).
Nothing here needs /root/reference at test time."""
import glob
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
TIED = sorted(glob.glob(os.path.join(GOLDEN, "ahc_tied_*_43200.json")))


def _first_mismatch(z, stem):
    want = np.load(stem + "_pairs.npz")["pairs"]
    bad = np.nonzero((z[:, :2].astype(np.int32) != want).any(axis=1))[0]
    return None if bad.size == 0 else (int(bad[0]), z[bad[0]].tolist(), want[bad[0]].tolist())


def test_tied_digests_are_committed():
    assert {os.path.basename(p) for p in TIED} == {"ahc_tied_dup30_43200.json", "ahc_tied_grid64_43200.json", "ahc_tied_silence5_43200.json"}


@pytest.mark.parametrize("route", ["auto", "reference-order matrix-filtered", "reference-order matrix-free"])
@pytest.mark.parametrize("path", TIED, ids=[os.path.basename(p)[9:-5] for p in TIED])
def test_tied_input_at_full_size_equals_the_reference_build(fa, gpu_ctx, switch, path, route):
    from ahc_full_inputs import ahc_tied_input, dendrogram_digest, sha256
    want = json.load(open(path))
    x = ahc_tied_input(want["dist"][len("tied_"):])
    assert x.shape == (want["n"], want["d"]) and sha256(x) == want["input_sha256"], "the seeded input did not regenerate bit-for-bit (numpy version?)"
    if route.endswith("matrix-free"):
        switch("FA_AHC_RO_NO_MATRIX", "1")
    mode = fa.AHC_MODE_AUTO if route == "auto" else fa.AHC_MODE_REFERENCE_ORDER
    st, z, stats = fa.linkage(x, mode=mode, ctx=gpu_ctx, return_stats=True)
    assert st == 0, gpu_ctx.last_error()
    assert stats["reference_order"] == 1, stats                 # AUTO met an exact tie at a minimum and took the tie route: these inputs are what they claim
    got = dendrogram_digest(z)
    if got["dendrogram_sha256"] != want["dendrogram_sha256"]:
        pytest.fail(f"dendrogram differs from the reference build: pairs equal {got['pairs_sha256'] == want['pairs_sha256']}, heights equal "
                    f"{got['heights_sha256'] == want['heights_sha256']}, first differing merge {_first_mismatch(z, path[:-5])}, stats {stats}")
    for thr, c in want["cuts"].items():
        lab = fa.cut(z, want["n"], float(thr))
        assert int(lab.max()) + 1 == c["clusters"]
        assert sha256(lab.astype(np.int32)) == c["labels_sha256"], f"labels at thr {thr}"
    print(f"{os.path.basename(path)} {route}: row for row the reference build; {stats}")
