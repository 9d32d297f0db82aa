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


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_walk_on_joint_logits_equals_walk_on_tables(fa, gpu_ctx, oracle_mod, dtype):
    """fa_tdt_greedy_logits_dev: the joint decisions (first-index argmax over the token logits, its softmax probability,
    first-index argmax over the duration logits) are taken inside the walk, on the visited rows only — the same result as
    building the decision tables for the whole (u, t) grid first (numpy) and walking those (CPU restatement)."""
    import torch
    rng = np.random.default_rng(11)
    B, U, T, V1, nd = 24, 40, 50, 130, 5
    blank = V1 - 1
    lg = rng.standard_normal((B, U, T, V1 + nd)).astype(np.float32)
    lg[..., blank] += 3.2                                  # ~75 % blanks
    lg[3, 0, 0, 5] = lg[3, 0, 0, 9] = 9.0                  # exact tie -> lowest index
    lg[4, 0, 1, :V1] = np.nan                              # NaN never wins: an all-NaN row decodes to token 0, probability 0
    lg[5, 0, :4, 7] = np.nan                              # one NaN among finite logits: the token is the finite argmax, the probability 0 after the clamp
    lg[6, 0, :4, :V1] = -np.inf                          # nothing above -inf: token 0, probability 0
    lg[7, 0, :4, 3:90] = -np.inf                          # -inf entries contribute nothing to the denominator
    lg = lg.astype(dtype)
    x = lg.astype(np.float32)
    tok = np.argmax(np.where(np.isnan(x[..., :V1]), -np.inf, x[..., :V1]), axis=-1).astype(np.int32)
    bn = np.argmax(x[..., V1:], axis=-1).astype(np.int32)
    mx = np.max(np.where(np.isnan(x[..., :V1]), -np.inf, x[..., :V1]), axis=-1, keepdims=True)
    with np.errstate(invalid="ignore"):
        pr = (1.0 / np.exp(x[..., :V1].astype(np.float64) - mx).sum(-1)).astype(np.float32)
    assert tok[3, 0, 0] == 5 and tok[4, 0, 1] == 0
    enc = rng.integers(T // 2, T + 1, B).astype(np.int32)
    t0 = rng.integers(0, 4, B).astype(np.int32)
    last = (rng.random(B) < 0.5).astype(np.int32)
    cfg = fa.TdtConfig(blank_id=blank)
    got = fa.tdt_decode_logits(torch.from_numpy(lg).cuda(), V1, enc, None, t0, last, None, None, config=cfg, max_out=256, ctx=gpu_ctx)
    for b in range(B):
        ref = oracle_mod.tdt_greedy(tok[b], bn[b], pr[b], enc[b], None, t0[b], bool(last[b]), 0, None, max_out=256, blank_id=blank)
        g = got[b]
        assert (g["status"], g["count"], g["final_u"], g["final_time"]) == (ref["status"], ref["count"], ref["final_u"], ref["final_time"]), b
        for k in ("tokens", "timestamps", "durations"):
            np.testing.assert_array_equal(g[k], ref[k])
        np.testing.assert_allclose(g["confidences"], ref["confidences"], rtol=2e-5, atol=1e-7)
    assert sum(g["count"] for g in got) > B


@pytest.mark.parametrize("dtype", ["float32", "float16"])
@pytest.mark.parametrize("V1", [5, 64, 130, 1025, 1087, 1088, 1089, 2500])   # fp16 rows of even stride (5, 1025, 1087 + 5 durations) take the pair requests
def test_logits_walk_every_row_form_and_every_option(fa, gpu_ctx, oracle_mod, dtype, V1):
    """Round 5: the walk on joint logits is a state machine with one joint evaluation per iteration (tdt_walk_wave); rows of up to 17 x 64 logits
    are decided from registers (row maximum + first index holding it, soft-max only when the token is emitted: tdt_logits_fits_kernel), longer
    rows in one streaming pass.  Every row form (1 .. 17 pieces, the 17th piece partly filled, both sides of the 1 088-logit boundary) with every
    per-chunk option of the entry — audio_frames, start frames beyond the chunk, last-chunk flush, global offsets, emit_after, a joint grid too
    small for the walk (OUTPUT_TOO_SMALL), room for fewer tokens than the walk emits — against the CPU restatement on tables built by numpy."""
    import torch
    rng = np.random.default_rng(V1)
    B, U, T, nd = 40, 14, 36, 5
    blank = V1 - 1
    lg = rng.standard_normal((B, U, T, V1 + nd)).astype(np.float32)
    lg[..., blank] += 2.0 + 0.25 * np.log2(V1)            # ~70 % blanks at every row length
    lg[3, 0, 0, 0] = lg[3, 0, 0, min(4, V1 - 2)] = 30.0    # exact tie -> lowest index
    lg[4, 0, 1, :V1] = np.nan                              # all NaN -> token 0, probability 0
    lg[5, 0, :4, min(2, V1 - 2)] = np.nan                  # one NaN among finite logits
    lg[6, 0, :4, :V1] = -np.inf                          # nothing above -inf -> token 0
    lg[7, 0, :3, V1 - 2] = 40.0                            # the maximum in the row's last-but-one slot (the clamped duplicates sit behind it)
    lg[8, 0, :3, V1:] = -np.inf                           # duration logits all -inf -> bin 0
    lg[9, 0, :3, V1 + 1] = np.nan                          # a NaN duration logit never wins
    lg = lg.astype(dtype)
    x = lg.astype(np.float32)
    xt = np.where(np.isnan(x[..., :V1]), -np.inf, x[..., :V1])
    tok = np.argmax(xt, axis=-1).astype(np.int32)
    xd = np.where(np.isnan(x[..., V1:]), -np.inf, x[..., V1:])
    bn = np.argmax(xd, axis=-1).astype(np.int32)
    mx = np.max(xt, axis=-1, keepdims=True)
    with np.errstate(invalid="ignore", over="ignore"):
        pr = (1.0 / np.exp(x[..., :V1].astype(np.float64) - mx).sum(-1)).astype(np.float32)
    enc = rng.integers(T // 2, T + 1, B).astype(np.int32)
    enc[0] = 1
    af = np.minimum(enc, rng.integers(T // 2, T + 4, B)).astype(np.int32)
    t0 = rng.integers(0, 5, B).astype(np.int32)
    t0[1] = T + 2
    last = (rng.random(B) < 0.5).astype(np.int32)
    goff = rng.integers(0, 300, B).astype(np.int32)
    ea = [None if rng.random() < 0.6 else int(goff[b] + rng.integers(0, 12)) for b in range(B)]
    cfg = fa.TdtConfig(blank_id=blank)
    for max_out in (64, 1):                                 # 1: room for fewer tokens than the walk emits (OUTPUT_TOO_SMALL, counts keep counting)
        got = fa.tdt_decode_logits(torch.from_numpy(lg).cuda(), V1, enc, af, t0, last, goff, ea, config=cfg, max_out=max_out, ctx=gpu_ctx)
        statuses = set()
        for b in range(B):
            ref = oracle_mod.tdt_greedy(tok[b], bn[b], pr[b], enc[b], af[b], t0[b], bool(last[b]), goff[b], ea[b], max_out=max_out, blank_id=blank)
            g = got[b]
            statuses.add(ref["status"])
            assert (g["status"], g["count"], g["final_u"], g["final_time"]) == (ref["status"], ref["count"], ref["final_u"], ref["final_time"]), (b, max_out)
            for k in ("tokens", "timestamps", "durations"):
                np.testing.assert_array_equal(g[k], ref[k])
            np.testing.assert_allclose(g["confidences"], ref["confidences"], rtol=3e-5, atol=1e-7)
        assert 0 in statuses
        if max_out == 64:
            assert sum(g["count"] for g in got) > B // 2        # the rows do produce tokens
