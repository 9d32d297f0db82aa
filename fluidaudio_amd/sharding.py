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


# ---- VBx sharded over the frame axis (reference loop: VBxClustering.swift:301-661) ------------------------------
VBX_SLICES = 64   # == fa_vbx_shard_slices(): everything that crosses frames is one record per slice


def vbx_shard_frames(T_total: int, rank: int, world_size: int) -> tuple[int, int]:
    """Frames [lo, hi) a rank holds (== fa_vbx_shard_range): 64 / world consecutive slices of ceil(T / 64) frames."""
    if world_size < 1 or VBX_SLICES % world_size or not (0 <= rank < world_size):
        raise ValueError("the world size must divide 64")
    per = -(-T_total // VBX_SLICES)
    zn = VBX_SLICES // world_size
    return min(rank * zn * per, T_total), min((rank + 1) * zn * per, T_total)


class VbxShard:
    """One rank's part of a sharded VBx run on its GPU (fa_vbx_shard_* of the C ABI).  `rho_local` / `labels_local` are the frames
    vbx_shard_frames() names, `n_speakers` the number of distinct labels of the WHOLE problem.  Records travel as torch CUDA tensors
    (the collective library moves them); begin() / iterate() return this rank's chunk, finish() the ELBO of the iteration."""

    def __init__(self, rho_local, labels_local, T_total: int, n_speakers: int, phi, rank: int, world_size: int,
                 warm_start_fa: float = 0.07, warm_start_fb: float = 0.8, ctx=None):
        import ctypes as C
        import torch
        from . import _lib as L
        self._L, self._C, self._torch = L, C, torch
        self.ctx = ctx or L.default_context()
        rho = np.ascontiguousarray(rho_local, np.float64)
        lab = np.ascontiguousarray(labels_local, np.int32)
        lo, hi = vbx_shard_frames(T_total, rank, world_size)
        D = rho.shape[1]
        if rho.shape[0] != hi - lo or lab.size != hi - lo:
            raise ValueError(f"rank {rank} of {world_size} holds the frames [{lo}, {hi}) of {T_total}")
        phi = np.ascontiguousarray(phi, np.float64)
        if phi.size != D:
            phi = np.ones(D)                                   # dimension mismatch -> identity (:72-76)
        self.frames, self.S, self.D, self.world = (lo, hi), int(n_speakers), D, world_size
        h = C.c_void_p()
        self.ctx.check(L.lib().fa_vbx_shard_create(self.ctx.handle, rho.ctypes.data, T_total, D, lab.ctypes.data, self.S, phi.ctypes.data,
                                                   warm_start_fa, warm_start_fb, rank, world_size, C.byref(h)), "fa_vbx_shard_create")
        self._h = h
        dev = torch.device("cuda", self.ctx.device)
        self._chunk = torch.empty(L.lib().fa_vbx_shard_chunk_doubles(self.S, D, world_size), dtype=torch.float64, device=dev)

    def begin(self):
        self.ctx.check(self._L.lib().fa_vbx_shard_begin(self._h, self._chunk.data_ptr()), "fa_vbx_shard_begin")
        return self._chunk

    def iterate(self, full):
        assert full.is_cuda and full.dtype == self._torch.float64 and full.numel() == self._chunk.numel() * self.world
        self._torch.cuda.current_stream(full.device).synchronize()   # the gather ran on torch's stream, the kernels run on the context's
        self.ctx.check(self._L.lib().fa_vbx_shard_iterate(self._h, full.data_ptr(), self._chunk.data_ptr()), "fa_vbx_shard_iterate")
        return self._chunk

    def finish(self, full) -> float:
        self._torch.cuda.current_stream(full.device).synchronize()
        e = self._C.c_double()
        self.ctx.check(self._L.lib().fa_vbx_shard_finish_iteration(self._h, full.data_ptr(), self._C.byref(e)), "fa_vbx_shard_finish_iteration")
        return e.value

    def result(self):
        n = self.frames[1] - self.frames[0]
        gamma, pi, hard = np.zeros((n, self.S)), np.zeros(self.S), np.zeros(n, np.int32)
        self.ctx.check(self._L.lib().fa_vbx_shard_result(self._h, gamma.ctypes.data, pi.ctypes.data, hard.ctypes.data), "fa_vbx_shard_result")
        return gamma, pi, hard

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.lib().fa_vbx_shard_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown: the library may already be gone)
            pass


def all_gather_records(dist=None):
    """gather(chunk) -> the chunks of all ranks in rank order, as one tensor (torch.distributed: "nccl" = RCCL on the GPU box)."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: F811

    def gather(chunk):
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
            return chunk
        if chunk.is_cuda and dist.get_backend() != "nccl":      # a CPU transport (gloo in the tests): staged through the host
            host = chunk.cpu()
            parts = [torch.empty_like(host) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, host)
            return torch.cat(parts).to(chunk.device)
        parts = [torch.empty_like(chunk) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, chunk)
        return torch.cat(parts)
    return gather


def vbx_refine_sharded(shard, gather, max_iterations: int = 20, convergence_tolerance: float = 1e-4):
    """The iteration loop of VBxClustering.runVBx (:301-661) over a sharded frame axis: ONE all-gather of the slice records per
    iteration; every rank evaluates the ELBO from the same records and stops in the same iteration (:653-659).
    `shard`: begin() / iterate(full) / finish(full) / result() (VbxShard on a GPU); returns (gamma_local, pi, hard_local, elbos)."""
    full = gather(shard.begin())
    prev, elbos = -np.inf, []
    for it in range(max_iterations):
        full = gather(shard.iterate(full))
        elbo = shard.finish(full)
        elbos.append(elbo)
        if it > 0 and abs(elbo - prev) < convergence_tolerance:
            break
        prev = elbo
    gamma, pi, hard = shard.result()
    return gamma, pi, hard, elbos
