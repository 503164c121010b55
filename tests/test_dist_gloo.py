"""world_size-2 CPU test (gloo) of the window sharding used for N > 1 GPUs: scatter from the
root, per-rank compute, gather of fixed-shape results.  The per-rank compute here is the CPU
oracle (test infrastructure); on the GPU box the same functions run with backend nccl (RCCL)
and the HIP engine."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyperiod_amd.dist import gather_rows, run_sharded, run_sharded_pipelined, scatter_windows, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_everything():
    for total in (1, 5, 8, 1000, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert max(hi - lo for lo, hi in spans) == -(-total // world)


def _worker(rank, world, port, total, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import period_oracle as po
        from pyperiod_amd.synth import multi_sinusoid_batch

        x_root = torch.from_numpy(multi_sinusoid_batch(0, total, n)) if rank == 0 else None
        local = scatter_windows(x_root, total, n, torch.float64, torch.device("cpu"))
        lo, hi = shard_bounds(total, world, rank)
        assert local.shape == (hi - lo, n)
        assert np.array_equal(local.numpy(), multi_sinusoid_batch(lo, hi - lo, n))
        back = gather_rows(local, total)
        if rank == 0:
            assert np.array_equal(back.numpy(), multi_sinusoid_batch(0, total, n))

        def compute(xl):  # stand-in for engine.m_best on this rank's windows
            per, pw = [], []
            for row in xl.numpy():
                p, w, _ = po.m_best(row, 3)
                per.append(p.astype(np.int64))
                pw.append(w)
            # an empty block (trailing rank) yields (0, 3) results, like the engine does
            return (torch.from_numpy(np.array(per, dtype=np.int64).reshape(-1, 3)),
                    torch.from_numpy(np.array(pw, dtype=np.float64).reshape(-1, 3)))

        want = [po.m_best(row, 3) for row in multi_sinusoid_batch(0, total, n)] if rank == 0 else None
        for runner in (run_sharded, lambda *a: run_sharded_pipelined(*a, pieces=2), lambda *a: run_sharded_pipelined(*a, pieces=(1, 3))):
            res = runner(compute, x_root, total, n, torch.float64, torch.device("cpu"))
            if rank == 0:
                per, pw = res
                assert per.shape == (total, 3)
                assert np.array_equal(per.numpy(), np.array([w[0] for w in want]))
                assert np.allclose(pw.numpy(), np.array([w[1] for w in want]), rtol=1e-12)
            else:
                assert res is None
        if rank == 0:
            q.put("ok")
    finally:
        dist.destroy_process_group()


def _launch(world, total, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_scatter_compute_gather_world2():
    _launch(2, 5, 256)  # odd count: the last rank's block is shorter


def test_empty_trailing_shard_world4():
    """total=5 over 4 ranks -> blocks of 2, 2, 1, 0 windows: the rank with the empty block must
    still take part in every collective (it used to raise before the gather and hang its peers)."""
    assert shard_bounds(5, 4, 3) == (5, 5)
    _launch(4, 5, 192)


def test_piece_rows():
    from pyperiod_amd.dist import piece_rows

    assert piece_rows(8192, 4) == [2048] * 4
    assert piece_rows(8192, (1, 3)) == [2048, 6144]
    assert piece_rows(8192, 1) == [8192] and piece_rows(8192, (1,)) == [8192]
    assert piece_rows(3, (1, 3)) == [1, 2] and piece_rows(2, (1, 3)) == [2] and piece_rows(5, 3) == [2, 2, 1]
    assert piece_rows(0, (1, 3)) == [0] and piece_rows(1, 4) == [1]
    for per in (1, 7, 100, 8191):
        for pc in (1, 2, 5, (1, 3), (1, 1, 2), (0.5, 0.25, 0.25)):
            assert sum(piece_rows(per, pc)) == per
