#!/usr/bin/env python3
"""AHC probe on cuda:0: wall-clock, device stats and (optionally) bit-exactness against the reference build.
usage: ahc_probe.py N[,N...] [--kinds iid,mix] [--modes 0,1] [--check MAXN]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidaudio_amd as fa  # noqa: E402
import oracle  # noqa: E402
from tests.conftest import speaker_mixture  # noqa: E402


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    ns = [int(v) for v in sys.argv[1].split(",")]
    kinds = arg("--kinds", "iid,mix").split(",")
    modes = [int(v) for v in arg("--modes", "0").split(",")]
    check = int(arg("--check", "10000"))
    ctx = fa.default_context(0)

    def run(x, mode):
        n, d = x.shape
        dx = torch.from_numpy(x).cuda()
        dz = torch.zeros((n - 1, 4), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        st = fa._lib.AhcStats()
        t = time.perf_counter()
        rc = fa.lib().fa_ahc_linkage(ctx.handle, C.c_void_p(dx.data_ptr()), n, d, C.c_void_p(dz.data_ptr()), (n - 1) * 4, mode, 1, C.byref(st))
        return rc, time.perf_counter() - t, st.as_dict(), dz.cpu().numpy()

    for n in ns:
        for kind in kinds:
            if kind == "iid":
                x = oracle.ahc_normalize(np.random.default_rng(0).standard_normal((n, 256)))
            elif kind == "e2e":      # the bench session (tests/golden/e2e_inputs.py), n = 5400 x hours
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
                from e2e_inputs import e2e_session
                x = oracle.ahc_normalize(e2e_session(n / 5400.0)["emb"].astype(np.float64))
            else:
                x = speaker_mixture(n, 256, 64, 0.02, 0)
            for mode in modes:
                if mode == 1 and n > 20000:
                    continue
                rc, t, s, z = run(x, mode)
                rc, t, s, z = run(x, mode)
                ok = tref = None
                if n <= check and mode == 0:
                    t0 = time.perf_counter()
                    sr, zr = oracle.linkage_ref(x)
                    tref = time.perf_counter() - t0
                    ok = bool(np.array_equal(z, zr))
                print(json.dumps(dict(n=n, kind=kind, mode=mode, rc=rc, wall_s=round(t, 4), ref_s=tref, bit_exact=ok,
                                      **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items()},
                                      us_per_round=round(1e3 * s["merge_ms"] / max(1, s["rounds"]), 2))), flush=True)


if __name__ == "__main__":
    main()
