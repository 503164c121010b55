#!/usr/bin/env python3
"""Benchmark of the pyPeriod projection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 -- BASELINE.json config 2, the configuration the metric is quoted on: one *step* =
  Periods.m_best(num=10) over 1024 synthetic windows x N=4096 fp64 resident in HBM: the step-1
  kernel (repeated all-p sweep p = 2..N/3, argmax, subtract -- one launch per window batch) and
  the step-2 factor-refinement kernel.  A window-projection is one (window, candidate period)
  projection + norm of an all-p sweep; winner re-projections and step-2
  projections are performed but NOT counted.  The line also carries `c4_single_gpu`: BASELINE
  config 4's whole batch (65 536 windows, small_to_large(0.05)) on this one GPU -- the
  denominator of the strong-scaling target below.

N > 1 -- BASELINE.json config 4 / BASELINE.md's scaling target, STRONG scaling: the whole batch
  of 65 536 windows x N=4096 fp64 starts on rank 0; one step = RCCL scatter (pipelined in pieces,
  pyperiod_amd/dist.py) -> small_to_large(0.05) on every rank's block -> RCCL gather of
  counts / periods / powers to rank 0.  `value` = 65 536 x 2047 nominal window-projections per
  step / wall time of the step (max over ranks), i.e. it includes the collectives.  Reported
  beside it: the unpipelined phase times (scatter / compute / gather) and the same 65 536
  windows on rank 0's GPU alone (`single_gpu`), so `speedup_vs_single_gpu` is the strong-scaling
  ratio measured inside one run.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline     -- dominant kernel, launch time from HIP events on the kernel's own stream.
                  The window lives in LDS, so the roof that binds is LDS read bandwidth
                  (~150 TB/s aggregate ds_read_b64), not HBM: `achieved` = bytes the fold passes
                  read from LDS per second (pass plan x N x 8 B, exact), `frac` <= 1.
                  `logical_hbm_ratio` is SURVEY 8d's logical figure (N*8 B per window-projection
                  / 8 TB/s; exceeds 1 because fusion works), `hbm_frac_measured` the rocprofv3
                  HBM bytes per launch (`traffic`, recorded in profiles/ by a separate PMC run)
                  / launch time / 8 TB/s.
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
WINDOWS_PER_GPU = 1024  # config 2
NUM_PERIODS = 10
C4_WINDOWS = 65536  # config 4
C4_THRESH = 0.05
C4_CAP = 32  # accepted periods kept per window (15 on average, max 24 on this data); overflow is checked
BYTES_PER_WINDOW_PROJECTION = N_SAMPLES * 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
LDS_PEAK_GBS = 150000.0  # MI355X_MICROARCH.md LDS: ~150 TB/s aggregate ds_read_b64


# ------------------------------------------------------------------------------------------
# host-side legs that must run BEFORE this process touches the GPU (they spawn workers)
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


def cpu_baseline(n, num, per_worker=64):
    """The oracle on the host cores of this box (one window per call, like the reference): 64 windows per core as
    BASELINE.md plans, about 16 s of CPU work."""
    import multiprocessing as mp

    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    cores = max(1, min(avail, 16))  # a one-GPU box owns a 16-core share of its host, whatever cpu_count says
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
        "os_cpu_count": os.cpu_count(),
        "affinity_cpus": avail,
        "kind": "port",
        "sample": f"oracle m_best(num={num}) on windows 0..{cores * per_worker - 1} (N={n}), "
        f"{per_worker} per core on {cores} cores, {projections} projections, {busy:.1f} s busy / {wall:.1f} s wall",
        "per_core": projections / busy / cores,
        "reference_per_core_survey": 6178.0,  # the reference itself, 1 Xeon core (different machine), BASELINE.md
    }


def _synth_worker(args):
    from pyperiod_amd.synth import multi_sinusoid_batch

    w0, count, n = args
    return w0, multi_sinusoid_batch(w0, count, n)


def synth_windows(total, n, procs=None):
    """Windows 0..total-1 of the seeded generator, built by a pool of host processes."""
    import multiprocessing as mp

    import numpy as np

    if procs is None:
        try:
            procs = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            procs = os.cpu_count() or 1
        procs = max(1, min(procs, 16))
    out = np.empty((total, n), dtype=np.float64)
    chunk = 1024
    jobs = [(w0, min(chunk, total - w0), n) for w0 in range(0, total, chunk)]
    if procs == 1:
        for job in jobs:
            w0, blk = _synth_worker(job)
            out[w0 : w0 + blk.shape[0]] = blk
        return out
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        for w0, blk in pool.imap_unordered(_synth_worker, jobs):
            out[w0 : w0 + blk.shape[0]] = blk
    return out


def load_recorded_traffic(kernel):
    """HBM bytes per launch of `kernel` measured with rocprofv3 --pmc (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes, + WRITE_SIZE) in a separate profiling run; profiles/traffic.json."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        ent = rec.get("kernels", {}).get(kernel)
        if ent:
            return ent.get("hbm_bytes_per_launch"), ent.get("source")
    except (OSError, ValueError):
        pass
    return None, None


def c2_workload():
    return f"config 2: m_best(num={NUM_PERIODS}) all-p sweep p=2..{N_SAMPLES // 3}, {WINDOWS_PER_GPU} windows x N={N_SAMPLES} fp64 per GPU"


def c4_workload(world, total=C4_WINDOWS):
    return (f"config 4: small_to_large(thresh={C4_THRESH}) p=2..{N_SAMPLES // 2}, {total} windows x N={N_SAMPLES} fp64, "
            f"batch on rank 0 -> RCCL scatter -> compute on {world} GPU(s) -> RCCL gather")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~0.9 s timed at N=1: long enough for a 1 Hz utilisation sampler
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="N=1: skip the config-4 single-GPU leg")
    ap.add_argument("--c4-windows", type=int, default=C4_WINDOWS, help="rehearsals only; the reported config is 65536")
    ap.add_argument("--pieces", type=int, default=8, help="N>1: pieces the scatter of each rank's block is cut into")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # ---- host-only legs first: nothing below this block may fork/spawn after a HIP call
    cpu = None
    x4_host = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(N_SAMPLES, NUM_PERIODS)
        if world > 1 or not args.no_c4:
            x4_host = synth_windows(args.c4_windows, N_SAMPLES)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    ndev = max(1, torch.cuda.device_count())  # does not initialise the GPU
    dev_index = local_rank % ndev  # == local_rank on a full node
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the process group is created before any other GPU work of this process
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(dev_index)

    import __graft_entry__ as ge

    ge.build()
    from pyperiod_amd import PeriodEngine
    from pyperiod_amd.synth import multi_sinusoid_batch

    eng = PeriodEngine(dev_index)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def kernel_ms(prof, name):
        v = [ms for nm, ms in prof if nm == name]
        return sum(v) / max(1, len(v)), len(v)

    def c4_compute(xl):
        counts, per, pw, _, st = eng.small_to_large(xl, C4_THRESH, None, False, False, cap=C4_CAP, want_bases=False,
                                                    nosync=True)
        return counts, per, pw, st

    line = None
    if world == 1:
        # =============================== config 2 (headline) ===============================
        x = torch.from_numpy(multi_sinusoid_batch(0, WINDOWS_PER_GPU, N_SAMPLES)).to(dev)
        P = N_SAMPLES // 3 - 2 + 1
        n_pass, n_per = eng.sweep_plan_info(2, N_SAMPLES // 3)
        assert n_per == P

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
        n_sweeps = int(sweeps.sum().item())
        proj = n_sweeps * P  # per step
        k1_ms, _ = kernel_ms(prof, "k_mbest_step1")
        k2_ms, _ = kernel_ms(prof, "k_mbest_step2")
        logical = proj * BYTES_PER_WINDOW_PROJECTION  # SURVEY 8d, per launch
        lds_bytes = n_sweeps * n_pass * BYTES_PER_WINDOW_PROJECTION  # what the fold passes read from LDS
        traffic, tsrc = load_recorded_traffic("k_mbest_step1")
        sec = k1_ms * 1e-3
        line = {
            "metric": "window-projections/sec (all-p sweep, N=4096)",
            "value": proj * args.steps / elapsed,
            "unit": "window-projections/s",
            "n_gpus": 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (3 sinusoids + 5 % noise, numpy default_rng(1000 + w), SURVEY 8d)",
            "config": {
                "workload": c2_workload(),
                "windows_per_gpu": WINDOWS_PER_GPU,
                "n_samples": N_SAMPLES,
                "periods_swept": P,
                "passes_per_sweep": n_pass,
                "sweeps_per_window_mean": n_sweeps / WINDOWS_PER_GPU,
                "parallelism": "one process per GPU; windows are independent, no data-path collective at N=1",
            },
            "roofline": {
                "kernel": "k_mbest_step1<double, true>",
                "bound": "lds",
                "achieved": lds_bytes / sec / 1e9 if sec > 0 else 0.0,
                "peak": LDS_PEAK_GBS,
                "unit": "GB/s",
                "frac": lds_bytes / sec / 1e9 / LDS_PEAK_GBS if sec > 0 else 0.0,
                "achieved_definition": "bytes the fold passes read from LDS per launch (sweeps x passes_per_sweep x N x 8 B; "
                "one pass yields up to 3 periods) / launch time",
                "launch_ms": k1_ms,
                "step2_launch_ms": k2_ms,
                "lds_bytes_per_launch": lds_bytes,
                "algorithmic_bytes_per_launch": logical,
                "logical_GBs": logical / sec / 1e9 if sec > 0 else 0.0,
                "logical_hbm_ratio": logical / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
                "logical_lds_ratio": logical / sec / 1e9 / LDS_PEAK_GBS if sec > 0 else 0.0,
                "traffic": traffic,
                "traffic_recorded": True,
                "traffic_source": tsrc,
                "hbm_frac_measured": (traffic / sec / 1e9 / HBM_PEAK_GBS) if (traffic and sec > 0) else None,
                # step 1 reads the windows and hands the basis rows to step 2 in compact form (the first p <= N/3
                # elements of each row); step 2 writes the (num, N) matrix once (its launch: step2_launch_ms)
                "compulsory_hbm_bytes_per_launch": WINDOWS_PER_GPU * N_SAMPLES * 8 + WINDOWS_PER_GPU * NUM_PERIODS * (N_SAMPLES // 3) * 8,
                "compulsory_hbm_bytes_step2": WINDOWS_PER_GPU * NUM_PERIODS * (N_SAMPLES // 3 + N_SAMPLES) * 8,
                "note": "SURVEY 8d's logical figure (N*8 B per window-projection against 8 TB/s) is kept as logical_hbm_ratio; it "
                "exceeds 1 because the fused sweep serves every pass but the first from LDS.  The binding roof is LDS read "
                "bandwidth / VALU issue; frac is the LDS utilisation.",
            },
            "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = line["value"] / cpu["value"]
        del x, out, periods, powers, bases
        # =============================== config 4 on this one GPU ===============================
        if x4_host is not None:
            x4 = torch.from_numpy(x4_host).to(dev)
            c4_compute(x4[:1024])
            torch.cuda.synchronize(dev)
            eng.profile(True)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                o4 = c4_compute(x4)
            torch.cuda.synchronize(dev)
            ms4 = 1e3 * (time.perf_counter() - t0) / reps
            k4_ms, _ = kernel_ms(eng.profile_read(), "k_small_to_large")
            eng.profile(False)
            assert int(o4[3].abs().max().item()) == 0, "small_to_large: capacity exceeded"
            units = x4.shape[0] * (N_SAMPLES // 2 - 1)
            line["c4_single_gpu"] = {
                "workload": f"config 4 on one GPU: small_to_large({C4_THRESH}), {x4.shape[0]} windows x N={N_SAMPLES}, resident in HBM",
                "ms": ms4,
                "kernel_ms": k4_ms,
                "window_projections_per_s": units / (ms4 * 1e-3),
                "accepted_periods_mean": float(o4[0].double().mean().item()),
                "logical_lds_ratio": units * BYTES_PER_WINDOW_PROJECTION / (k4_ms * 1e-3) / 1e9 / LDS_PEAK_GBS if k4_ms else None,
            }
    else:
        # =============================== config 4, strong scaling ===============================
        from pyperiod_amd.dist import gather_rows, run_sharded_pipelined, scatter_windows

        total = args.c4_windows
        x_root = torch.from_numpy(x4_host).to(dev) if rank == 0 else None
        del x4_host

        def step():
            return run_sharded_pipelined(c4_compute, x_root, total, N_SAMPLES, torch.float64, dev, pieces=args.pieces)

        for _ in range(args.warmup):
            res = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        barrier()
        elapsed = time.perf_counter() - t0
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

        # phase breakdown, unpipelined, outside the timed region (host clock, max over ranks)
        barrier()
        t1 = time.perf_counter()
        xl = scatter_windows(x_root, total, N_SAMPLES, torch.float64, dev)
        barrier()
        t2 = time.perf_counter()
        eng.profile(True)
        o = c4_compute(xl)
        barrier()
        t3 = time.perf_counter()
        k_ms, _ = kernel_ms(eng.profile_read(), "k_small_to_large")
        eng.profile(False)
        g = [gather_rows(t, total) for t in o]
        barrier()
        t4 = time.perf_counter()
        kmax = torch.tensor([k_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)

        single = None
        checks = {}
        if rank == 0:
            # consistency checks are REPORTED, not asserted: a failing rank must not leave its peers in a collective
            counts, per, pw, st = res
            checks["shapes"] = bool(counts.shape == (total,) and per.shape == (total, C4_CAP) and pw.shape == (total, C4_CAP))
            checks["capacity_ok"] = bool(int(st.abs().max().item()) == 0)
            checks["pipelined_equals_unpipelined"] = bool(torch.equal(counts, g[0]) and torch.equal(per, g[1]))
            c4_compute(x_root[:1024])
            torch.cuda.synchronize(dev)
            reps = 3
            ts = time.perf_counter()
            for _ in range(reps):
                o1 = c4_compute(x_root)
            torch.cuda.synchronize(dev)
            single_ms = 1e3 * (time.perf_counter() - ts) / reps
            checks["sharded_equals_single_gpu"] = bool(torch.equal(o1[0], counts) and torch.equal(o1[1], per))
            single = {"ms": single_ms, "window_projections_per_s": total * (N_SAMPLES // 2 - 1) / (single_ms * 1e-3),
                      "note": "the same batch on rank 0's GPU alone, input resident, no collective"}
        barrier()
        if rank == 0:
            units = total * (N_SAMPLES // 2 - 1)  # nominal projections per step
            ms_per_step = 1e3 * elapsed / args.steps
            per_gpu_units = -(-total // world) * (N_SAMPLES // 2 - 1)
            kms = float(kmax.item())
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
            except Exception:  # noqa: BLE001
                rccl = None
            line = {
                "metric": "window-projections/sec (all-p sweep, N=4096)",
                "value": units * args.steps / elapsed,
                "unit": "window-projections/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": ms_per_step,
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": "f64",
                "data": "synthetic (3 sinusoids + 5 % noise, numpy default_rng(1000 + w), SURVEY 8d)",
                "config": {
                    "workload": c4_workload(world, total),
                    "total_windows": total,
                    "n_samples": N_SAMPLES,
                    "periods_swept": N_SAMPLES // 2 - 1,
                    "scatter_pieces": args.pieces,
                    "parallelism": f"contiguous blocks of {-(-total // world)} windows per rank; scatter + gather only, no all-reduce",
                    "backend": args.backend,
                    "world_size": dist.get_world_size(),
                    "rccl_version": rccl,
                },
                "phases_unpipelined_ms": {"scatter": 1e3 * (t2 - t1), "compute": 1e3 * (t3 - t2), "gather": 1e3 * (t4 - t3),
                                          "sum": 1e3 * (t4 - t1)},
                "single_gpu": single,
                "speedup_vs_single_gpu": single["ms"] / ms_per_step,
                "checks": checks,
                "roofline": {
                    "kernel": "k_small_to_large<double, true>",
                    "bound": "lds",
                    "achieved": per_gpu_units * BYTES_PER_WINDOW_PROJECTION / (kms * 1e-3) / 1e9 if kms else 0.0,
                    "peak": LDS_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": per_gpu_units * BYTES_PER_WINDOW_PROJECTION / (kms * 1e-3) / 1e9 / LDS_PEAK_GBS if kms else 0.0,
                    "achieved_definition": "nominal screen passes (one per candidate period) x N x 8 B read from LDS per launch / "
                    "launch time of the slowest rank; re-screens after an accepted period are not counted",
                    "launch_ms": kms,
                    "traffic": (load_recorded_traffic("k_small_to_large")[0] if -(-total // world) == 8192 else None),
                    "traffic_recorded": True,
                    "traffic_source": load_recorded_traffic("k_small_to_large")[1],
                    "logical_hbm_ratio": per_gpu_units * BYTES_PER_WINDOW_PROJECTION / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else 0.0,
                },
                "cpu_baseline": None,
            }
    if line is not None:
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
