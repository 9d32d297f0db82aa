#!/usr/bin/env python3
"""call 23 aborted inside test_slots_per_thread_forms_equal_reference_build at the tied input: the same sequence outside pytest (stderr visible)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluidaudio_amd as fa
import oracle
ctx = fa.default_context()
for cpt in ("1", "2", "4"):
    rng = np.random.default_rng(int(cpt))
    cases = [oracle.ahc_normalize(rng.standard_normal((n, d))) for n, d in ((2, 3), (3, 5), (257, 16), (513, 32), (700, 64), (1024, 8), (1500, 24))]
    dup = oracle.ahc_normalize(rng.standard_normal((400, 8)))
    dup = np.concatenate([dup, dup[:150]])
    for single_block in (True, False):
        os.environ["FA_AHC_CPT"] = cpt
        if single_block: os.environ.pop("FA_AHC_NO_SINGLE_BLOCK", None)
        else: os.environ["FA_AHC_NO_SINGLE_BLOCK"] = "1"
        for mode in (fa.AHC_MODE_AUTO, fa.AHC_MODE_EXACT):
            for x in cases:
                st, z = fa.linkage(x, ctx=ctx, mode=mode)
                assert st == 0
        print("cpt", cpt, "single_block", single_block, "-> tied input", flush=True)
        for form in ("", "1"):
            if form: os.environ["FA_AHC_RO_NO_MATRIX"] = "1"
            else: os.environ.pop("FA_AHC_RO_NO_MATRIX", None)
            st, z, stats = fa.linkage(dup, ctx=ctx, return_stats=True)
            print("   form", form or "matrix", "status", st, stats["reference_order"], "equal ref", bool(np.array_equal(z, oracle.linkage_ref(dup)[1])), flush=True)
        os.environ.pop("FA_AHC_RO_NO_MATRIX", None)
        bad = cases[4].copy(); bad[300, 5] = np.nan
        print("   nan status", fa.linkage(bad, ctx=ctx)[0], flush=True)
print("done")
