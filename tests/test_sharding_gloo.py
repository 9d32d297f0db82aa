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
