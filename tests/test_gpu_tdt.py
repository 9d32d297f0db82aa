"""TDT greedy control loop on the device vs the CPU restatement, over random joint-decision tables."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,U,T,p_blank,seed", [(64, 40, 60, 0.7, 0), (200, 160, 188, 0.85, 1), (32, 12, 30, 0.3, 2)])
def test_batched_walk_matches_oracle(fa, gpu_ctx, oracle_mod, B, U, T, p_blank, seed):
    import torch
    rng = np.random.default_rng(seed)
    blank = 8192
    tok = rng.integers(0, 50, (B, U, T)).astype(np.int32)
    tok[rng.random((B, U, T)) < p_blank] = blank
    bn = rng.integers(0, 5, (B, U, T)).astype(np.int32)
    pr = rng.uniform(-0.2, 1.2, (B, U, T)).astype(np.float32)
    pr[rng.random((B, U, T)) < 0.01] = np.nan
    enc = rng.integers(max(2, T // 2), T + 1, B).astype(np.int32)
    enc[0] = 1                                           # guard (:110-112)
    af = np.minimum(enc, rng.integers(T // 2, T + 5, B)).astype(np.int32)
    t0 = rng.integers(0, 8, B).astype(np.int32)
    t0[1] = T + 3                                        # starts beyond the chunk (:150-152)
    last = (rng.random(B) < 0.4).astype(np.int32)
    goff = rng.integers(0, 400, B).astype(np.int32)
    ea = [None if rng.random() < 0.7 else int(goff[b] + rng.integers(0, 20)) for b in range(B)]
    bn[2, 0, :] = 6                                      # duration bin out of range -> per-chunk error status
    got = fa.tdt_decode_tables(torch.from_numpy(tok).cuda(), torch.from_numpy(bn).cuda(), torch.from_numpy(pr).cuda(), enc, af, t0,
                               last, goff, ea, max_out=256, ctx=gpu_ctx)
    for b in range(B):
        ref = oracle_mod.tdt_greedy(tok[b], bn[b], pr[b], enc[b], af[b], t0[b], bool(last[b]), goff[b], ea[b], max_out=256)
        g = got[b]
        assert g["status"] == ref["status"], b
        assert g["count"] == ref["count"] and g["final_u"] == ref["final_u"] and g["final_time"] == ref["final_time"], b
        for k in ("tokens", "timestamps", "durations"):
            np.testing.assert_array_equal(g[k], ref[k])
        np.testing.assert_array_equal(g["confidences"], ref["confidences"])
    assert got[0]["final_time"] is None and got[1]["final_time"] is None and got[2]["status"] == 5
    assert sum(g["count"] for g in got) > B   # the tables do produce tokens
