"""Two ranks on ONE GPU, backend gloo (collectives staged through the host), the HIP engine as the per-rank compute:
the sharded runners of pyperiod_amd/dist.py must give the single-process results bit for bit.

Not a test module: tests/conftest.py starts `python tests/dist_gpu_job.py --launch OUT.json` as a child process at
session start -- before the pytest process itself has touched the GPU -- and tests/test_gpu_dist.py reads OUT.json.
The ranks are fresh processes started by the launcher, which never imports torch.
"""
import json
import os
import socket
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def launch(out_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(WORLD):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", out_path], env=env))
    deadline = time.time() + 600
    rc = 0
    live = list(procs)
    stop = []
    import signal

    signal.signal(signal.SIGTERM, lambda *_: stop.append(1))  # the session ended early: end the ranks, then leave
    while live and not rc and not stop and time.time() < deadline:
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    for p in live:
        p.kill()
        p.wait()
        rc = rc or 1
    if rc and not os.path.exists(out_path):
        with open(out_path, "w") as fh:
            json.dump({"ok": False, "error": f"a rank exited with {rc}"}, fh)
    sys.exit(rc)


def rank_main(out_path):
    sys.path.insert(0, ROOT)
    rank = int(os.environ["RANK"])
    res = {"ok": False}
    try:
        import numpy as np
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
        from pyperiod_amd import PeriodEngine
        from pyperiod_amd.dist import run_sharded, run_sharded_pipelined
        from pyperiod_amd.synth import multi_sinusoid_batch

        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        eng = PeriodEngine(0)
        checks = {}

        def s2l(xl):
            c, p, w, _, st = eng.small_to_large(xl, 0.05, None, False, False, cap=32, want_bases=False, nosync=True)
            return c, p, w, st

        def mbest(xl):
            return eng.m_best(xl, 6)

        def qo(xl):
            return eng.qo_find_periods(xl, 3, 0.1, 8, 200, 512)

        for tag, fn, total, n, dtype, npdt in (("small_to_large", s2l, 301, 2048, torch.float64, np.float64),
                                               ("m_best", mbest, 37, 1536, torch.float64, np.float64),
                                               ("qo_find_periods", qo, 21, 4096, torch.float32, np.float32)):
            x_root = torch.from_numpy(multi_sinusoid_batch(0, total, n, dtype=npdt)).to(dev) if rank == 0 else None
            for rname, runner in (("sharded", run_sharded), ("pipelined", lambda *a: run_sharded_pipelined(*a, pieces=3))):
                got = runner(fn, x_root, total, n, dtype, dev)
                if rank == 0:
                    want = fn(x_root)
                    checks[f"{tag}_{rname}"] = bool(len(got) == len(want) and all(
                        g.shape == w.shape and torch.equal(g, w) for g, w in zip(got, want)))
                else:
                    assert got is None
            dist.barrier()
        res = {"ok": all(checks.values()), "checks": checks, "world": dist.get_world_size(), "backend": dist.get_backend()}
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        res = {"ok": False, "error": traceback.format_exc()}
        if rank == 0:
            with open(out_path, "w") as fh:
                json.dump(res, fh)
        raise
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(res, fh)


if __name__ == "__main__":
    if sys.argv[1] == "--launch":
        launch(sys.argv[2])
    else:
        rank_main(sys.argv[2])
