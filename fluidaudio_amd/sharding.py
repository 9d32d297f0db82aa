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
