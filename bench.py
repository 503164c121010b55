#!/usr/bin/env python3
"""Headline benchmark: window-projections/s of the fused all-p sweep, BASELINE.json config 2.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = Periods.m_best(num=10) over one batch of 1024 synthetic windows x N=4096 fp64 per
GPU, already resident in HBM: the step-1 kernel (repeated all-p sweep p = 2..N/3, argmax,
subtract -- one launch per window batch) and the step-2 factor-refinement kernel.  A
window-projection is one (window, candidate period) projection + norm of an all-p sweep
(SURVEY.md 8d); winner re-projections and step-2 projections are performed but NOT counted,
so the figure is conservative.  Windows are independent, so ranks never exchange data:
"scaling" is weak (1024 windows per GPU).

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline     -- step-1 kernel: algorithmic bytes (32768 B per window-projection)
                  / launch time from HIP events on the kernel's own stream, against 8 TB/s.
                  The window lives in LDS, so this logical figure may exceed the HBM peak;
                  `lds_frac` (vs ~150 TB/s of LDS read bandwidth) is the utilisation that
                  actually binds, `traffic` the measured HBM bytes per launch (rocprofv3 PMC,
                  profiles/) when available.
  cpu_baseline -- the numpy oracle (a port of the reference) timed on the host cores of this
                  box on a bounded sample of the same windows (N=1 only).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = 4096
WINDOWS_PER_GPU = 1024
NUM_PERIODS = 10
BYTES_PER_WINDOW_PROJECTION = N_SAMPLES * 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
LDS_PEAK_GBS = 150000.0  # MI355X_MICROARCH.md LDS: ~150 TB/s aggregate ds_read_b64


# ------------------------------------------------------------------------------------------
# CPU baseline leg: the oracle, one window per call like the reference.  Runs in spawned
# worker processes BEFORE this process touches the GPU.
# ------------------------------------------------------------------------------------------
def _cpu_worker(args):
    w0, count, n, num = args
    import numpy as np  # noqa: F401

    from oracle import period_oracle as po
    from pyperiod_amd.synth import multi_sinusoid_window

    calls = [0]
    inner = po.project

    def counting_project(*a, **k):
        calls[0] += 1
        return inner(*a, **k)

    po.project = counting_project
    windows = [multi_sinusoid_window(w0 + i, n) for i in range(count)]
    t0 = time.perf_counter()
    for x in windows:
        po.m_best(x, num)
    dt = time.perf_counter() - t0
    # the winner's base is taken from the sweep, so every call outside step 2 is a sweep entry
    return calls[0], dt


def cpu_baseline(n, num, per_worker=6):
    import multiprocessing as mp

    cores = max(1, min(os.cpu_count() or 1, 16))
    ctx = mp.get_context("spawn")
    jobs = [(i * per_worker, per_worker, n, num) for i in range(cores)]
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    projections = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {
        "value": projections / busy,
        "unit": "window-projections/s",
        "cores": cores,
        "kind": "port",
        "sample": f"oracle m_best(num={num}) on windows 0..{cores * per_worker - 1} (N={n}), "
        f"{per_worker} per core, {projections} projections, {busy:.1f} s busy / {wall:.1f} s wall",
        "per_core": projections / busy / cores,
    }


def load_recorded_traffic():
    """HBM bytes per step-1 launch measured with rocprofv3 --pmc (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes, + WRITE_SIZE); recorded in profiles/ by the profiling run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        if rec.get("workload") == workload_name():
            return rec.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def workload_name():
    return f"m_best(num={NUM_PERIODS}) all-p sweep p=2..{N_SAMPLES // 3}, {WINDOWS_PER_GPU} windows x N={N_SAMPLES} fp64 per GPU"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--root-leg", action="store_true",
                    help="N > 1: also time the batch-on-rank-0 deployment shape (RCCL scatter -> m_best -> gather), "
                         "reported separately, never part of `value`")
    ap.add_argument("--no-root-leg", action="store_true", help=argparse.SUPPRESS)  # accepted, the leg is opt-in now
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(N_SAMPLES, NUM_PERIODS)  # before any HIP call in this process

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge

    ge.build()
    from pyperiod_amd import PeriodEngine
    from pyperiod_amd.synth import multi_sinusoid_batch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = local_rank % max(1, torch.cuda.device_count())  # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    eng = PeriodEngine(dev_index)
    # windows rank*1024 .. rank*1024+1023 of the seeded generator; resident in HBM before t0
    x_host = multi_sinusoid_batch(rank * WINDOWS_PER_GPU, WINDOWS_PER_GPU, N_SAMPLES)
    x = torch.from_numpy(x_host).to(dev)
    P = N_SAMPLES // 3 - 2 + 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        return eng.m_best(x, NUM_PERIODS, None, 2, False, want_sweeps=True)

    for _ in range(args.warmup):
        out = step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)

    periods, powers, bases, status, sweeps = out
    assert int(status.abs().sum().item()) == 0, "a window failed in m_best"

    # Optional second leg (N > 1 only, outside the timed region): the batch starts on rank 0,
    # is scattered with RCCL, processed, and the fixed-shape results are gathered back --
    # the "RCCL scatter/gather over xGMI" deployment shape.  Never part of `value`.
    root_leg = None
    if world > 1 and args.root_leg and not args.no_root_leg:
        try:
            from pyperiod_amd.dist import gather_rows, scatter_windows

            total = world * WINDOWS_PER_GPU
            x_root = torch.cat([x] * world, 0) if rank == 0 else None  # synthetic: rank 0's windows repeated
            barrier()
            t1 = time.perf_counter()
            xl = scatter_windows(x_root, total, N_SAMPLES, torch.float64, dev)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            o = eng.m_best(xl, NUM_PERIODS, None, 2, False)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            g_per = gather_rows(o[0], total)
            g_pow = gather_rows(o[1], total)
            barrier()
            t4 = time.perf_counter()
            root_leg = {"scatter_ms": 1e3 * (t2 - t1), "compute_ms": 1e3 * (t3 - t2), "gather_ms": 1e3 * (t4 - t3),
                        "total_ms": 1e3 * (t4 - t1), "windows": total,
                        "note": "input on rank 0 -> RCCL scatter -> m_best -> RCCL gather of periods/powers"}
            if rank == 0:
                assert g_per.shape == (total, NUM_PERIODS) and g_pow.shape == (total, NUM_PERIODS)
        except Exception as exc:  # the headline number must survive a failure of this leg
            root_leg = {"error": repr(exc)}
    proj_local = int(sweeps.sum().item()) * P  # per step on this rank
    tproj = torch.tensor([float(proj_local)], device=dev, dtype=torch.float64)
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tproj, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    proj_total = float(tproj.item())
    elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = proj_total * args.steps / elapsed
        k1 = [ms for name, ms in prof if name == "k_mbest_step1"]
        k2 = [ms for name, ms in prof if name == "k_mbest_step2"]
        k1_ms = sum(k1) / max(1, len(k1))
        alg_bytes = proj_local * BYTES_PER_WINDOW_PROJECTION
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        line = {
            "metric": "window-projections/sec (all-p sweep, N=4096)",
            "value": value,
            "unit": "window-projections/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload_name(),
                "windows_per_gpu": WINDOWS_PER_GPU,
                "n_samples": N_SAMPLES,
                "periods_swept": P,
                "sweeps_per_window_mean": float(sweeps.double().mean().item()),
                "parallelism": f"windows sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "kernel": "k_mbest_step1<double, true>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_recorded_traffic(),
                "algorithmic_bytes_per_launch": alg_bytes,
                "launch_ms": k1_ms,
                "step2_launch_ms": sum(k2) / max(1, len(k2)),
                "lds_frac": achieved / LDS_PEAK_GBS,
                "note": "logical bytes (N*8 per window-projection); the window is LDS-resident, so frac can exceed 1 -- lds_frac is the binding utilisation",
            },
            "cpu_baseline": cpu,
        }
        if root_leg is not None:
            line["rccl_root_leg"] = root_leg
        if cpu:
            line["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
