"""GPU parity: VBx refinement (HIP fp64) vs the CPU oracle.  The reference has no numeric test of runVBx
(parity unpinned beyond the restatement); tolerance: 1e-9 absolute on gamma/pi, 1e-9 relative on the ELBO history,
identical iteration counts and hard assignments."""
import numpy as np
import pytest
from test_oracle_vbx import make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,D,K,seed", [(600, 32, 5, 0), (2500, 128, 7, 1), (64, 16, 2, 2), (1000, 128, 70, 3)])
def test_vbx_matches_oracle(fa, gpu_ctx, oracle_mod, T, D, K, seed):
    x, init, phi = make_problem(T, D, K, seed)
    gamma, pi, hard, elbos = oracle_mod.vbx_refine(x, init, phi)
    out = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    assert out.num_clusters == K and len(out.elbos) == len(elbos)
    np.testing.assert_allclose(out.elbos, elbos, rtol=1e-9)
    np.testing.assert_allclose(out.gamma, gamma, atol=1e-9)
    np.testing.assert_allclose(out.pi, pi, atol=1e-9)
    assert out.hard_clusters[0] == hard.tolist()
    np.testing.assert_allclose(out.gamma.sum(1), 1.0, atol=1e-12)
    assert np.all(np.diff(out.elbos) > -1e-6)


def test_vbx_edge_cases(fa, gpu_ctx, oracle_mod):
    v = fa.VBxClustering(np.ones(8), ctx=gpu_ctx)
    out = v.refine(np.zeros((0, 8)), [])
    assert out.num_clusters == 0 and out.hard_clusters == []
    x, init, phi = make_problem(200, 8, 3, 5)
    init[:] = 4  # one distinct label -> S = 1, labels clamp to speaker 0 (:78, :104)
    out = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    assert out.num_clusters == 1 and set(out.hard_clusters[0]) == {0}
    # deterministic: two runs are bit-identical
    x, init, phi = make_problem(800, 64, 6, 9)
    a = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    b = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    assert np.array_equal(a.gamma, b.gamma) and a.elbos == b.elbos


@pytest.mark.parametrize("T,D,K,seed,world", [(2500, 128, 7, 1, 2), (2500, 128, 7, 1, 8), (1000, 128, 70, 3, 4), (50, 16, 3, 4, 8), (43200, 128, 12, 6, 8)])
def test_vbx_sharded_over_frames_equals_single_device_bit_for_bit(fa, gpu_ctx, oracle_mod, T, D, K, seed, world):
    """SURVEY §8(e) row 4 (VBxClustering.swift:301-661 sharded over T): `world` shards of 64 / world slices each — here all on one GPU,
    the all-gather being a torch.cat — reproduce fa_vbx_refine bit for bit: gamma, pi, hard labels, every ELBO, the iteration count.
    (T = 50 < 64 slices: most shards hold no frame at all.)  The N > 1 transport (torch.distributed all_gather) is covered by the gloo test."""
    import torch
    from fluidaudio_amd.sharding import VbxShard, vbx_refine_sharded, vbx_shard_frames
    x, init, phi = make_problem(T, D, K, seed)
    one = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    S = one.num_clusters
    shards = []
    for r in range(world):
        lo, hi = vbx_shard_frames(T, r, world)
        shards.append(VbxShard(x[lo:hi], init[lo:hi], T, S, phi, r, world, ctx=gpu_ctx))

    class Lockstep:     # all ranks of the job in one process: every call runs on each shard, the "collective" is a concatenation
        def begin(self): return [s.begin().clone() for s in shards]
        def iterate(self, full): return [s.iterate(full).clone() for s in shards]
        def finish(self, full):
            e = [s.finish(full) for s in shards]
            assert all(v == e[0] for v in e)        # every rank computes the same ELBO from the same records
            return e[0]
        def result(self):
            parts = [s.result() for s in shards]
            assert all(np.array_equal(p[1], parts[0][1]) for p in parts)
            return np.concatenate([p[0] for p in parts]), parts[0][1], np.concatenate([p[2] for p in parts])

    gamma, pi, hard, elbos = vbx_refine_sharded(Lockstep(), lambda chunks: torch.cat(chunks), 20, 1e-4)
    for s in shards:
        s.close()
    assert elbos == one.elbos
    assert np.array_equal(gamma, one.gamma) and np.array_equal(pi, one.pi) and hard.tolist() == one.hard_clusters[0]
    if T <= 2500:
        og, op, oh, oe = oracle_mod.vbx_refine(x, init, phi)
        assert len(oe) == len(elbos) and np.array_equal(oh, hard)
        np.testing.assert_allclose(gamma, og, atol=1e-9)


def test_vbx_shard_argument_checks(fa, gpu_ctx):
    from fluidaudio_amd.sharding import VbxShard, vbx_shard_frames
    with pytest.raises(ValueError):
        vbx_shard_frames(100, 0, 3)                     # 3 does not divide 64
    x, init, phi = make_problem(200, 8, 3, 5)
    with pytest.raises(ValueError):
        VbxShard(x[:10], init[:10], 200, 3, phi, 0, 2, ctx=gpu_ctx)   # not the frames this rank holds


def _vbx_rank(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    import fluidaudio_amd as fa
    from fluidaudio_amd.sharding import VbxShard, all_gather_records, vbx_refine_sharded, vbx_shard_frames
    from test_oracle_vbx import make_problem
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, init, phi = make_problem(3000, 64, 6, 8)
    lo, hi = vbx_shard_frames(len(x), rank, world)
    shard = VbxShard(x[lo:hi], init[lo:hi], len(x), 6, phi, rank, world, ctx=fa.default_context())
    gamma, pi, hard, elbos = vbx_refine_sharded(shard, all_gather_records(dist), 20, 1e-4)
    shard.close()
    q.put((rank, gamma, pi, hard, elbos))
    dist.barrier()
    dist.destroy_process_group()


def test_vbx_sharded_two_processes_one_gpu(fa, gpu_ctx):
    """The N > 1 path end to end with real processes: two ranks (both on this box's one GPU, each with its own context), the device
    kernels of fa_vbx_shard_*, torch.distributed.all_gather for the records (gloo here, staged through the host; "nccl" = RCCL on a
    multi-GPU node) — gamma, pi, hard labels and every ELBO equal the single-device fa_vbx_refine bit for bit."""
    import socket
    import torch.multiprocessing as mp
    from test_oracle_vbx import make_problem
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_vbx_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x, init, phi = make_problem(3000, 64, 6, 8)
    one = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    assert got[0][4] == got[1][4] == one.elbos
    assert np.array_equal(np.concatenate([got[0][1], got[1][1]]), one.gamma)
    assert np.array_equal(got[0][2], one.pi) and np.array_equal(got[1][2], one.pi)
    assert np.concatenate([got[0][3], got[1][3]]).tolist() == one.hard_clusters[0]


@pytest.mark.parametrize("T,D,K,seed", [(3000, 128, 70, 3), (5000, 128, 300, 4), (700, 40, 49, 5), (4097, 96, 129, 6)])
def test_tiled_kernels_of_many_speakers_keep_the_bits(fa, gpu_ctx, oracle_mod, switch, T, D, K, seed):
    """S >= 48 speakers (hard sessions: hundreds of AHC clusters): the two contractions of an iteration run as 64 x 64 tiled products
    (vbx_gt_rho_tiled, vbx_logits_tiled + vbx_softmax_rows).  Every output is one accumulator fed in ascending k by the same fused
    multiply-adds as the one-speaker-per-wavefront kernels: gamma, pi, ELBOs and labels are identical bit for bit (FA_VBX_NO_TILED=1 runs
    the old kernels), and equal to the CPU restatement at its tolerance."""
    x, init, phi = make_problem(T, D, K, seed)
    tiled = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    switch("FA_VBX_NO_TILED", "1")
    plain = fa.VBxClustering(phi, ctx=gpu_ctx).refine(x, init)
    switch("FA_VBX_NO_TILED", None)
    assert tiled.num_clusters == plain.num_clusters == K
    assert tiled.elbos == plain.elbos
    np.testing.assert_array_equal(tiled.gamma, plain.gamma)
    np.testing.assert_array_equal(tiled.pi, plain.pi)
    assert tiled.hard_clusters == plain.hard_clusters
    if T * K <= 400_000:
        gamma, pi, hard, elbos = oracle_mod.vbx_refine(x, init, phi)
        np.testing.assert_allclose(tiled.elbos, elbos, rtol=1e-9)
        np.testing.assert_allclose(tiled.gamma, gamma, atol=1e-9)
        assert tiled.hard_clusters[0] == hard.tolist()
