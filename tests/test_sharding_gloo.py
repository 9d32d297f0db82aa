"""world_size-2 gloo test of the N>1 path: ranks own contiguous unit slices, results gather on rank 0 in unit order."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fluidaudio_amd.sharding import gather_ragged_int32, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_units = 11
    lo, hi = shard_range(n_units, rank, world)
    rows = [np.arange(u % 4, dtype=np.int32) + 100 * u for u in range(lo, hi)]  # ragged, some empty
    out = gather_ragged_int32(rows, dist, dst=0)
    # weak-scaling bookkeeping of bench.py: max-over-ranks of a per-rank time
    import torch
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put(([r.tolist() for r in out], float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_in_unit_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert rows == [(np.arange(u % 4) + 100 * u).tolist() for u in range(11)]
    assert tmax == 2.0


def _worker_minima(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fluidaudio_amd.sharding import row_minima_numpy, row_minima_sharded, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((173, 24))                      # odd N: slabs of 87 and 86 rows
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[40] = x[12]                                           # an exact tie (distance 0) across the slab boundary's side
    lo, hi = shard_range(len(x), rank, world)
    m, a = row_minima_sharded(x[lo:hi], lo, len(x), row_minima_numpy, dist)
    if rank == 1:                                           # every rank holds the full table
        q.put((m.tolist(), a.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_partition_reproduces_single_rank_row_minima():
    """the sharded start-up of one linkage problem (all-gather X, per-rank row slab, all-gather of (min, idx))
    gives exactly the single-rank nearest-neighbour table, ties included."""
    from fluidaudio_amd.sharding import row_minima_numpy
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_minima, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    m, a = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    x = rng.standard_normal((173, 24))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[40] = x[12]
    m1, a1 = row_minima_numpy(x, 0, len(x))
    assert a == a1.tolist() and m == m1.tolist()
    assert a[40] == 12 and a[12] == 40 and m[40] == 0.0


def _vbx_problem():
    rng = np.random.default_rng(11)
    T, D, K = 1000, 16, 5
    means = rng.standard_normal((K, D)) * 2.0
    spk = rng.integers(0, K, T)
    rho = means[spk] + 0.7 * rng.standard_normal((T, D))
    init = spk.copy()
    flip = rng.random(T) < 0.25                               # a quarter of the AHC labels are wrong: VBx has work to do
    init[flip] = rng.integers(0, K + 2, flip.sum())           # ... and two clusters nobody really uses
    phi = rng.random(D) * 3.0 + 0.1
    return rho, init.astype(np.int32), phi, len(np.unique(init))


def _worker_vbx(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from fluidaudio_amd.sharding import all_gather_records, vbx_refine_sharded, vbx_shard_frames
    from vbx_shard_numpy import NumpyVbxShard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rho, init, phi, S = _vbx_problem()
    lo, hi = vbx_shard_frames(len(rho), rank, world)
    shard = NumpyVbxShard(rho[lo:hi], init[lo:hi], len(rho), S, phi, rank, world)
    gamma, pi, hard, elbos = vbx_refine_sharded(shard, all_gather_records(dist), 20, 0.5)
    q.put((rank, lo, hi, gamma, pi, hard, elbos))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_vbx_over_frames_equals_single_rank_and_oracle():
    """VBx sharded over T (one all-gather of the 64 slice records per iteration) gives the single-rank run bit
    for bit, stops in the same iteration on both ranks, and agrees with the CPU restatement of VBxClustering.swift:167-664."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from fluidaudio_amd.sharding import vbx_refine_sharded
    from vbx_shard_numpy import NumpyVbxShard
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_vbx, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rho, init, phi, S = _vbx_problem()
    assert (got[0][1], got[0][2], got[1][1], got[1][2]) == (0, 512, 512, 1000)      # 32 slices of 16 frames each, the last ones short
    gamma = np.concatenate([got[0][3], got[1][3]])
    hard = np.concatenate([got[0][5], got[1][5]])
    assert got[0][6] == got[1][6] and np.array_equal(got[0][4], got[1][4])           # same ELBOs, same pi on both ranks
    one = NumpyVbxShard(rho, init, len(rho), S, phi, 0, 1)
    g1, p1, h1, e1 = vbx_refine_sharded(one, lambda c: c, 20, 0.5)
    assert np.array_equal(gamma, g1) and np.array_equal(got[0][4], p1) and np.array_equal(hard, h1) and got[0][6] == e1
    og, op, oh, oe = oracle.vbx_refine(rho, init, phi, 20, 0.5)   # a tolerance this problem reaches: the stopping rule is exercised
    assert len(oe) == len(e1) and 2 < len(e1) < 20
    np.testing.assert_allclose(e1, oe, rtol=1e-9)
    np.testing.assert_allclose(g1, og, rtol=0, atol=1e-9)
    np.testing.assert_allclose(p1, op, rtol=0, atol=1e-9)
    assert np.array_equal(h1, oh)
