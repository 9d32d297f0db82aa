"""Data-parallel sharding of the hot path across the GPUs of one node.

Utterance batches, logit matrices and independent recordings partition with NO data-path
collective: every rank featurizes / decodes / clusters its own contiguous
slice.  The only exchange is the optional gather of small results (token ids, lengths,
labels) on rank 0, which goes through ``torch.distributed`` (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_units: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [lo, hi) slice of `n_units` owned by `rank`; sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n_units, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_offsets(offsets, rank: int, world_size: int):
    """Slice a packed-utterance offsets array: returns (first_utt, local_offsets starting at 0, sample_lo, sample_hi)."""
    offsets = np.asarray(offsets, np.int64)
    lo, hi = shard_range(offsets.size - 1, rank, world_size)
    return lo, offsets[lo:hi + 1] - offsets[lo], int(offsets[lo]), int(offsets[hi])


def gather_ragged_int32(local_rows: list, dist=None, dst: int = 0):
    """Gather per-unit int32 rows (token ids / labels) from every rank onto `dst`, in unit order.

    One `gather` of lengths + one `gather` of a padded flat buffer: two small collectives per
    job, independent of the amount of audio processed."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: F811
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [np.asarray(r, np.int32) for r in local_rows]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    lens = torch.tensor([len(r) for r in local_rows], dtype=torch.int32, device=dev)
    meta = torch.tensor([lens.numel(), int(lens.sum())], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_units = max(int(m[0]) for m in metas)
    max_tok = max(int(m[1]) for m in metas)
    lens_p = torch.zeros(max_units, dtype=torch.int32, device=dev)
    lens_p[:lens.numel()] = lens
    flat = torch.zeros(max(max_tok, 1), dtype=torch.int32, device=dev)
    if len(local_rows):
        cat = np.concatenate([np.asarray(r, np.int32) for r in local_rows]) if int(lens.sum()) else np.zeros(0, np.int32)
        flat[:cat.size] = torch.from_numpy(cat).to(dev)
    g_lens = [torch.zeros_like(lens_p) for _ in range(world)] if rank == dst else None
    g_flat = [torch.zeros_like(flat) for _ in range(world)] if rank == dst else None
    dist.gather(lens_p, g_lens, dst=dst)
    dist.gather(flat, g_flat, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        n_units = int(metas[r][0])
        ls = g_lens[r][:n_units].cpu().numpy()
        fl = g_flat[r].cpu().numpy()
        pos = 0
        for n in ls:
            out.append(fl[pos:pos + n].copy())
            pos += int(n)
    return out


def row_minima_sharded(x_local_rows, row_lo: int, n_total: int, slab_fn, dist=None):
    """Start-up of ONE linkage problem sharded by rows (SURVEY.md §8e "AHC, one 50 k problem: partially"):

      1. all-gather of the unit-normalised embeddings X (N d fp64 bytes; ring over xGMI) — every rank then holds all of X;
      2. every rank computes the nearest-neighbour entries of ITS slab of rows [lo, hi) against all N points
         (``slab_fn(x_all, lo, hi) -> (mins, args)``: fa_ahc_row_minima on the GPU, a numpy restatement in the gloo test);
      3. all-gather of the (min, arg) pairs (12 N bytes) — every rank ends with the full table.

    x_local_rows: this rank's contiguous rows [row_lo, row_lo + len) of X.  Returns (mins [N] fp64, args [N] int32).
    The serial merge chain that follows does not shard (one 16-byte all-reduce per merge would cost more than the whole
    single-GPU merge phase), so the table is consumed by ONE rank's merge loop; see DESIGN.md §4."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: F811
    x_local = np.ascontiguousarray(x_local_rows, np.float64)
    single = not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1
    if single:
        assert row_lo == 0 and x_local.shape[0] == n_total
        m, a = slab_fn(x_local, 0, n_total)
        return np.asarray(m, np.float64), np.asarray(a, np.int32)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    d = x_local.shape[1]
    lo, hi = shard_range(n_total, rank, world)
    assert lo == row_lo and hi - lo == x_local.shape[0], "rows must follow shard_range"
    per = -(-n_total // world)                                   # padded slab (all_gather wants equal shapes)
    pad = torch.zeros((per, d), dtype=torch.float64, device=dev)
    pad[:hi - lo] = torch.from_numpy(x_local).to(dev)
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)                                  # 1. X everywhere
    x_all = np.concatenate([parts[r][:shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]].cpu().numpy() for r in range(world)])
    m, a = slab_fn(x_all, lo, hi)                                # 2. own slab
    mp = torch.full((per,), float("inf"), dtype=torch.float64, device=dev)
    ap = torch.full((per,), -1, dtype=torch.int32, device=dev)
    mp[:hi - lo] = torch.from_numpy(np.asarray(m, np.float64)).to(dev)
    ap[:hi - lo] = torch.from_numpy(np.asarray(a, np.int32)).to(dev)
    gm = [torch.zeros_like(mp) for _ in range(world)]
    ga = [torch.zeros_like(ap) for _ in range(world)]
    dist.all_gather(gm, mp)                                      # 3. the table everywhere
    dist.all_gather(ga, ap)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    return (np.concatenate([gm[r][:sizes[r]].cpu().numpy() for r in range(world)]),
            np.concatenate([ga[r][:sizes[r]].cpu().numpy() for r in range(world)]))


def row_minima_numpy(x_all, lo: int, hi: int):
    """CPU restatement of fa_ahc_row_minima for the gloo tests: squared distances as the sequential sum over the dimension
    (cumsum keeps the order), nearest other point with the lowest index on ties."""
    x = np.asarray(x_all, np.float64)
    mins, args = np.empty(hi - lo), np.empty(hi - lo, np.int32)
    for i in range(lo, hi):
        dist = np.cumsum((x[i][None, :] - x) ** 2, axis=1)[:, -1]
        dist[i] = np.inf
        j = int(np.argmin(dist))                                 # first minimum = lowest index
        mins[i - lo], args[i - lo] = dist[j], j
    return mins, args
