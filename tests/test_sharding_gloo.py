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
