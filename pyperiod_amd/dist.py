"""Window sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Signal windows are independent units -- no algorithm on this path exchanges data between
windows -- so the only collectives are the optional scatter of the input batch from a root
rank and the gather of the fixed-shape results back to it.  There is no all-reduce anywhere.
Ragged results (small_to_large) travel as fixed (W, cap) slabs plus a count per window.
"""

from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of ceil(total / world) windows for `rank`."""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def scatter_windows(x_root, total: int, n: int, dtype, device, src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` holds the (total, n) batch; every rank returns its own (hi-lo, n) block on
    `device`.  One scatter of equal ceil(total/world)-row chunks (the tail chunk is padded)."""
    world, rank = _world(group)
    lo, hi = shard_bounds(total, world, rank)
    if world == 1:
        return x_root.to(device)
    per = -(-total // world)
    recv = torch.empty((per, n), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        xr = x_root.to(device)
        if xr.shape[0] < per * world:
            pad = torch.zeros((per * world - xr.shape[0], n), dtype=dtype, device=device)
            xr = torch.cat([xr, pad], 0)
        chunks = [xr[r * per : (r + 1) * per].contiguous() for r in range(world)]
    dist.scatter(recv, chunks, src=src, group=group)
    return recv[: hi - lo]


def gather_rows(local: torch.Tensor, total: int, dst: int = 0, group=None):
    """Inverse of scatter_windows for any per-window result whose leading axis is the local
    window axis.  Returns the (total, ...) tensor on rank `dst`, None elsewhere."""
    world, rank = _world(group)
    if world == 1:
        return local
    per = -(-total // world)
    send = local
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], 0)
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(bufs, 0)[:total]


def run_sharded(fn: Callable[[torch.Tensor], Sequence[torch.Tensor]], x_root, total: int, n: int, dtype, device,
                root: int = 0, group=None):
    """scatter -> fn(local windows) -> gather.  `fn` returns a tuple of tensors whose first
    axis is the local window axis (e.g. ``lambda x: engine.m_best(x, 10)``).  Rank `root`
    gets the tuple of gathered (total, ...) tensors; other ranks get None."""
    x_local = scatter_windows(x_root, total, n, dtype, device, root, group)
    outs = fn(x_local)
    gathered = [gather_rows(o, total, root, group) for o in outs]
    _, rank = _world(group)
    return tuple(gathered) if rank == root else None
