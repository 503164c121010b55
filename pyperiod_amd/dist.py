"""Window sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Signal windows are independent units -- no algorithm on this path exchanges data between
windows -- so the only collectives are the scatter of the input batch from a root rank and the
gather of the fixed-shape results back to it.  There is no all-reduce anywhere.  Ragged results
(small_to_large) travel as fixed (W, cap) slabs plus a count per window.

Partition: contiguous blocks of ceil(total / world) windows per rank; trailing ranks
may get a short or an EMPTY block (total=5, world=4 -> 2, 2, 1, 0) and still take part in every
collective -- the engine returns empty outputs for an empty batch.

`run_sharded_pipelined` cuts every rank's block into pieces and scatters piece k+1 while piece k
is being processed: the root's xGMI links (7 x ~153 GB/s, point to point) carry 7/8 of the batch,
which for config 4 (2 GiB in, ~10 ms of compute per GPU) costs as much as the compute itself
unless the two overlap.
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


def _host_staged(group=None) -> bool:
    """gloo moves host memory only (scatter / gather have no GPU path there): device tensors are
    staged through the host.  Used by the CPU tests and by single-GPU rehearsals of the N > 1
    code path; with backend nccl (RCCL) device buffers go straight over xGMI."""
    return dist.get_backend(group) == "gloo"


def _rows(x_root: torch.Tensor, lo: int, count: int, total: int) -> torch.Tensor:
    """Rows [lo, lo+count) of the root batch, zero-padded past `total` (only the pieces that
    straddle the end of the batch are copied; the others are views)."""
    hi = min(total, lo + count)
    if hi - lo == count:
        return x_root[lo:hi]
    out = torch.zeros((count,) + tuple(x_root.shape[1:]), dtype=x_root.dtype, device=x_root.device)
    if hi > lo:
        out[: hi - lo] = x_root[lo:hi]
    return out


def scatter_windows(x_root, total: int, n: int, dtype, device, src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` holds the (total, n) batch; every rank returns its own (hi-lo, n) block on
    `device`.  One scatter of equal ceil(total/world)-row chunks (the tail chunk is padded)."""
    world, rank = _world(group)
    lo, hi = shard_bounds(total, world, rank)
    if world == 1:
        return x_root.to(device)
    per = -(-total // world)
    cdev = torch.device("cpu") if _host_staged(group) else device
    recv = torch.empty((per, n), dtype=dtype, device=cdev)
    chunks = None
    if rank == src:
        xr = x_root.to(cdev)
        chunks = [_rows(xr, r * per, per, total).contiguous() for r in range(world)]
    dist.scatter(recv, chunks, src=src, group=group)
    return recv[: hi - lo].to(device)


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
    if _host_staged(group):
        send = send.cpu()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(bufs, 0)[:total].to(local.device)


def run_sharded(fn: Callable[[torch.Tensor], Sequence[torch.Tensor]], x_root, total: int, n: int, dtype, device,
                root: int = 0, group=None):
    """scatter -> fn(local windows) -> gather.  `fn` returns a tuple of tensors whose first
    axis is the local window axis and whose trailing shape does not depend on the data (e.g.
    ``lambda x: engine.m_best(x, 10)``); it is also called for an empty block and must return
    (0, ...) tensors then (every engine method does).  Rank `root` gets the tuple of gathered
    (total, ...) tensors; other ranks get None."""
    x_local = scatter_windows(x_root, total, n, dtype, device, root, group)
    outs = fn(x_local)
    gathered = [gather_rows(o, total, root, group) for o in outs]
    _, rank = _world(group)
    return tuple(gathered) if rank == root else None


def piece_rows(per: int, pieces) -> list[int]:
    """Rows of the pieces a block of `per` windows is cut into.  `pieces` is a count (equal pieces) or a sequence of
    weights, e.g. (1, 3) = a first piece of a quarter of the block and one launch for the rest.  Every launch ends with a
    partly empty chip and a small piece is mostly tail, so fewer and larger pieces compute faster (tools/s2l_scale.py, one
    rank's 8192-window block of config 4: one launch of small_to_large 5.5 ms, two halves 5.55, a quarter + the rest
    6.0, four quarters 6.2-6.6); the first piece's scatter is the only one no kernel hides."""
    if isinstance(pieces, int):
        n = max(1, min(int(pieces), max(per, 1)))
        step = -(-per // n)
        rows = [min(step, per - k * step) for k in range(n)]
    else:
        w = [float(v) for v in pieces if float(v) > 0]
        tot = sum(w) or 1.0
        rows, used = [], 0
        for k, v in enumerate(w):
            r = per - used if k == len(w) - 1 else min(per - used, int(round(per * v / tot)))
            rows.append(r)
            used += r
    return [r for r in rows if r > 0] or [0]


def run_sharded_pipelined(fn: Callable[[torch.Tensor], Sequence[torch.Tensor]], x_root, total: int, n: int, dtype,
                          device, pieces=2, root: int = 0, group=None):
    """Like run_sharded, with the scatter cut into pieces (`piece_rows`) -- one asynchronous collective per piece and
    rank: all of them are enqueued up front (RCCL runs them on its own stream, in order), and `fn` runs on piece k as
    soon as it has landed while the later pieces are still in flight.  Results are concatenated per rank and gathered
    once per output tensor."""
    world, rank = _world(group)
    lo, hi = shard_bounds(total, world, rank)
    if world == 1:
        return tuple(fn(x_root.to(device)))
    per = -(-total // world)
    rows_of = piece_rows(per, pieces)
    cdev = torch.device("cpu") if _host_staged(group) else device
    xr = x_root.to(cdev) if rank == root else None
    recv, works, offs = [], [], []
    off = 0
    for rows in rows_of:
        buf = torch.empty((rows, n), dtype=dtype, device=cdev)
        lists = None
        if rank == root:
            lists = [_rows(xr, r * per + off, rows, total).contiguous() for r in range(world)]
        works.append(dist.scatter(buf, lists, src=root, group=group, async_op=True))
        recv.append(buf)
        offs.append(off)
        off += rows
    outs = []
    for buf, work, off in zip(recv, works, offs):
        work.wait()  # orders the current stream behind this piece only
        valid = max(0, min(hi - lo - off, buf.shape[0]))
        outs.append(fn(buf[:valid].to(device)))
    merged = [torch.cat([o[i] for o in outs], 0) for i in range(len(outs[0]))]
    gathered = [gather_rows(o, total, root, group) for o in merged]
    return tuple(gathered) if rank == root else None
