#!/usr/bin/env python3
"""Benchmark of the pyPeriod projection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own N ranks (fresh child processes,
before anything touches the GPU) and relays rank 0's line; under torch.distributed.run it joins the ranks it is given.

`value` is the SAME workload at every N (weak scaling): BASELINE.json config 2, the configuration the metric is
quoted on.  Every rank owns 1024 synthetic windows x N=4096 fp64 resident in HBM (rank r: windows r*1024 ...); one
*step* = Periods.m_best(num=10) over them: the step-1 kernel (repeated all-p sweep p = 2..N/3, argmax, subtract --
one launch per window batch) and the step-2 factor-refinement kernel.  A window-projection is one (window,
candidate period) projection + norm of an all-p sweep; winner re-projections and step-2 projections
are performed but NOT counted.  Windows are independent: there is no data-path collective; `value` = the
window-projections of all ranks / the slowest rank's time.

Beside it, under stable keys:
  c4            BASELINE config 4 / BASELINE.md's strong-scaling target: 65 536 windows, small_to_large(0.05).
                N = 1: the whole batch on the one GPU.  N > 1: the batch starts on rank 0; RCCL scatter (pipelined
                in pieces, pyperiod_amd/dist.py) -> compute on every rank's block -> RCCL gather of counts /
                periods / powers; the same batch on rank 0's GPU alone is timed in the same run
                (`speedup_vs_single_gpu`, wall and compute-only).  Consistency checks (sharded == single GPU, bit
                for bit) make the run exit non-zero when one fails.
  sweep_fp64    north_star's literal kernel (N = 1 only): the fused all-p projection sweep in fp64 -- ph_sweep / k_sweep, every
                ||P_p x|| for p = 2..N/3 of 1024 windows x N=4096 written out (Periods.py:501-510), no screen, no argmax.
  strong_scaling (N > 1) the headline figures of `c4` repeated at the top level: `value` itself is weak scaling of config 2
                with no collective and grows ~N x by construction -- it is NOT the scaling result and not comparable with the
                round-2 multi-GPU lines (which carried config 4 in `value`).
  c3_single_gpu BASELINE config 3 (N = 1 only): RamanujanPeriods.find_periods, 4096 windows x N=8192, q = 2..512.
  c5_single_gpu BASELINE config 5's per-GPU batch (N = 1 only): QOPeriods.find_periods, 1024 windows x N=16384 fp32.
  roofline      dominant kernel of `value`, launch time from HIP events on the kernel's own stream.  The window lives
                in LDS, so the roof that binds is LDS read bandwidth (~150 TB/s aggregate ds_read_b64) and VALU issue,
                not HBM: `achieved` = bytes the fold passes read from LDS per second (exact from the pass plan),
                `frac` <= 1.  `logical_lds_ratio` / `logical_hbm_ratio` are SURVEY 8d's algorithmic figure (N*8 B per
                window-projection) against the LDS peak / 8 TB/s; they exceed the physical figure because one pass
                yields up to three periods and -- since round 3 -- one 8-byte LDS element carries a float sample of
                TWO windows.  `traffic` = rocprofv3 HBM bytes per launch recorded by tools/profile_all.sh, null when
                the kernel sources have changed since.
  cpu_baseline  the numpy oracle (a port of the reference) timed on the host cores of this box on a bounded sample of
                the same windows (N = 1 only).
"""

import argparse
import hashlib
import json
import os
import subprocess
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
C3_WINDOWS, C3_N, C3_QHI = 4096, 8192, 512  # config 3
C5_WINDOWS, C5_N, C5_NUM, C5_THRESH, C5_MIN, C5_MAX, C5_KCAP = 1024, 16384, 3, 0.1, 8, 300, 1024  # config 5, one GPU's batch
BYTES_PER_WINDOW_PROJECTION = N_SAMPLES * 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
LDS_PEAK_GBS = 150000.0  # MI355X_MICROARCH.md LDS: ~150 TB/s aggregate ds_read_b64
METRIC = "window-projections/sec (all-p sweep, N=4096)"
DATA = "synthetic (3 sinusoids + 5 % noise, numpy default_rng(1000 + w), SURVEY 8d)"


# ------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` (N > 1) outside torch.distributed.run
# ------------------------------------------------------------------------------------------
def launch_ranks(n):
    """Start n fresh ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay their output and
    exit with the worst return code.  The parent never imports torch or touches the GPU."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_DEADLINE_S", "1500"))  # a hung rendezvous must not outlive the box's limit
    while live and not rc and time.time() < deadline:
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    if live and not rc:
        print(f"bench: ranks still running at the deadline, killing them", file=sys.stderr)
        rc = 1
    for p in live:  # a rank that died leaves its peers in a collective: end them (exact PIDs, our own children)
        p.kill()
        p.wait()
    sys.exit(rc if rc > 0 else (1 if rc else 0))


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


def _host_cores(limit=16):
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    return max(1, min(avail, limit)), avail  # a one-GPU box owns a 16-core share of its host, whatever cpu_count says


def _pool_map(fn, jobs, procs, unordered=False):
    """multiprocessing 'spawn' pool that is closed and joined (Pool.__exit__ terminates the workers, which a
    profiler's signal handler reports as 16 aborts)."""
    import multiprocessing as mp

    pool = mp.get_context("spawn").Pool(procs)
    try:
        out = list(pool.imap_unordered(fn, jobs) if unordered else pool.map(fn, jobs))
    finally:
        pool.close()
        pool.join()
    return out


def cpu_baseline(n, num, per_worker=64):
    """The oracle on the host cores of this box (one window per call, like the reference): 64 windows per core as
    BASELINE.md plans, about 16 s of CPU work."""
    cores, avail = _host_cores()
    jobs = [(i * per_worker, per_worker, n, num) for i in range(cores)]
    t0 = time.perf_counter()
    res = _pool_map(_cpu_worker, jobs, cores)
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
    import numpy as np

    from pyperiod_amd.synth import multi_sinusoid_batch

    w0, count, n, f32 = args
    return w0, multi_sinusoid_batch(w0, count, n, dtype=np.float32 if f32 else np.float64)


def synth_windows(total, n, f32=False, procs=None):
    """Windows 0..total-1 of the seeded generator, built by a pool of host processes."""
    import numpy as np

    if procs is None:
        procs = _host_cores()[0]
    out = np.empty((total, n), dtype=np.float32 if f32 else np.float64)
    chunk = max(64, min(1024, total // (2 * procs)))
    jobs = [(w0, min(chunk, total - w0), n, f32) for w0 in range(0, total, chunk)]
    if procs == 1:
        res = [_synth_worker(j) for j in jobs]
    else:
        res = _pool_map(_synth_worker, jobs, procs, unordered=True)
    for w0, blk in res:
        out[w0 : w0 + blk.shape[0]] = blk
    return out


def csrc_hash():
    """sha256 over the kernel sources: recorded PMC traffic is only quoted for the build it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyperiod_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def load_recorded_traffic(kernel):
    """HBM bytes per launch of `kernel` measured with rocprofv3 --pmc (FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes, + WRITE_SIZE) in a separate profiling run (tools/profile_all.sh -> profiles/traffic.json).
    -> (bytes, source, avg ms of that capture) or (None, reason, None) when the sources have changed since."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, "profiles/traffic.json missing", None
    ent = rec.get("kernels", {}).get(kernel)
    if not ent:
        return None, "no record for this kernel", None
    if ent.get("csrc_hash") != csrc_hash():
        return None, f"stale: recorded for csrc {ent.get('csrc_hash')}, this build is {csrc_hash()}", None
    return ent.get("hbm_bytes_per_launch"), ent.get("source"), ent.get("avg_ms_rocprof")


def c2_workload(world):
    return (f"config 2: m_best(num={NUM_PERIODS}) all-p sweep p=2..{N_SAMPLES // 3}, {WINDOWS_PER_GPU} windows x "
            f"N={N_SAMPLES} fp64 per GPU on {world} GPU(s), windows resident in HBM")


def c4_workload(world, total=C4_WINDOWS):
    return (f"config 4: small_to_large(thresh={C4_THRESH}) p=2..{N_SAMPLES // 2}, {total} windows x N={N_SAMPLES} fp64, "
            f"batch on rank 0 -> RCCL scatter -> compute on {world} GPU(s) -> RCCL gather")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~0.7 s timed at N=1: long enough for a 1 Hz utilisation sampler
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="N=1: skip the config-3 leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the config-4 leg")
    ap.add_argument("--no-c5", action="store_true", help="N=1: skip the config-5 leg")
    ap.add_argument("--c4-windows", type=int, default=C4_WINDOWS, help="rehearsals only; the reported config is 65536")
    ap.add_argument("--pieces", default="2", help="N>1: how the scatter of each rank's block is cut: a count of equal pieces "
                    "(default 2) or comma-separated weights, e.g. 1,3 (pyperiod_amd/dist.py piece_rows)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    pieces = int(args.pieces) if "," not in args.pieces else tuple(float(v) for v in args.pieces.split(","))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would describe a different job")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # ---- host-only legs first: nothing below this block may fork/spawn after a HIP call
    import __graft_entry__ as ge

    ge.build()  # hipcc runs as a child process: before the GPU is touched, before a communicator exists
    cpu = None
    x4_host = x3_host = x5_host = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(N_SAMPLES, NUM_PERIODS)
        if not args.no_c4:
            x4_host = synth_windows(args.c4_windows, N_SAMPLES)
        if world == 1 and not args.no_c3:
            x3_host = synth_windows(C3_WINDOWS, C3_N)
        if world == 1 and not args.no_c5:
            x5_host = synth_windows(C5_WINDOWS, C5_N, f32=True)
    from pyperiod_amd.synth import multi_sinusoid_batch

    x2_host = multi_sinusoid_batch(rank * WINDOWS_PER_GPU, WINDOWS_PER_GPU, N_SAMPLES)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    ndev = max(1, torch.cuda.device_count())  # does not initialise the GPU
    dev_index = local_rank % ndev  # == local_rank on a full node
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and world > torch.cuda.device_count():
            raise SystemExit(f"--gpus {world} with backend nccl (RCCL) needs {world} GPUs, this box shows {torch.cuda.device_count()}: "
                             "two ranks on one device cannot form a communicator (use --backend gloo for a rehearsal)")
        # the process group is created before any other GPU work of this process
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    torch.cuda.set_device(dev_index)

    from pyperiod_amd import PeriodEngine

    eng = PeriodEngine(dev_index)
    staged = world > 1 and args.backend == "gloo"  # gloo moves host tensors only

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def all_max(v):
        if world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cpu" if staged else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(v):
        if world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cpu" if staged else dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def kernel_ms(prof, name):
        v = [ms for nm, ms in prof if nm == name]
        return sum(v) / max(1, len(v)), len(v)

    WARM_S = 0.3

    def warm_clock(fn):
        """The secondary legs start behind host work (synthesis, copies), i.e. from an idle shader clock, and their few
        repetitions end before it has ramped (k_sweep: 0.41-0.47 ms from idle, 0.36 ms behind 0.2 s of load, on one box):
        they run their own call for WARM_S seconds first, as the main metric runs its warm-up steps."""
        t_end = time.perf_counter() + WARM_S
        while time.perf_counter() < t_end:
            fn()
            torch.cuda.synchronize(dev)

    def c4_compute(xl):
        counts, per, pw, _, st = eng.small_to_large(xl, C4_THRESH, None, False, False, cap=C4_CAP, want_bases=False,
                                                    nosync=True)
        return counts, per, pw, st

    checks = {}
    # =============================== config 2 (headline, every N) ===============================
    x = torch.from_numpy(x2_host).to(dev)
    P = N_SAMPLES // 3 - 2 + 1
    n_pass, n_per = eng.m_best_plan_info(N_SAMPLES, NUM_PERIODS)
    assert n_per == P
    win_per_wg, lds_elem = eng.m_best_info(N_SAMPLES, NUM_PERIODS)

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
    elapsed = all_max(time.perf_counter() - t0)
    prof = eng.profile_read()
    eng.profile(False)
    periods, powers, bases, status, sweeps = out
    checks["c2_all_windows_ok"] = all_sum(int(status.abs().sum().item())) == 0
    n_sweeps = int(sweeps.sum().item())
    proj = n_sweeps * P  # per step, this rank
    proj_all = all_sum(proj)
    k1_ms, _ = kernel_ms(prof, "k_mbest_step1")
    k2_ms, _ = kernel_ms(prof, "k_mbest_step2")
    k1_max = all_max(k1_ms)
    line = None
    if rank == 0:
        logical = proj * BYTES_PER_WINDOW_PROJECTION  # SURVEY 8d, per launch
        # fold passes of one launch: a workgroup sweeps until its last window is done
        sw = sweeps.cpu().numpy().astype(np.int64)
        if win_per_wg == 2:
            if sw.size % 2:
                sw = np.append(sw, 0)
            wg_sweeps = int(np.maximum(sw[0::2], sw[1::2]).sum())
        else:
            wg_sweeps = int(sw.sum())
        lds_bytes = wg_sweeps * n_pass * N_SAMPLES * lds_elem  # what the fold passes read from LDS
        traffic, tsrc, t_ms = load_recorded_traffic("k_mbest_step1_pair" if win_per_wg == 2 else "k_mbest_step1")
        sec = k1_ms * 1e-3
        line = {
            "metric": METRIC,
            "value": proj_all * args.steps / elapsed,
            "unit": "window-projections/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "value_note": None if world == 1 else "config 2 on every GPU, weak scaling, no data-path collective: ~N x by construction; "
                          "the strong-scaling result (config 4 over RCCL scatter / gather) is under `strong_scaling` / `c4`",
            "vs_baseline": None,
            "dtype": "f64",
            "data": DATA,
            "config": {
                "workload": c2_workload(world),
                "windows_per_gpu": WINDOWS_PER_GPU,
                "n_samples": N_SAMPLES,
                "periods_swept": P,
                "passes_per_sweep": n_pass,
                "sweeps_per_window_mean": n_sweeps / WINDOWS_PER_GPU,
                "step1_kernel": ("window-pair float screen + fp64 re-evaluation of the survivors (two windows per workgroup)"
                                 if win_per_wg == 2 else "one window per workgroup, fp64 fold"),
                "parallelism": "one process per GPU; windows are independent, no data-path collective",
                "backend": args.backend if world > 1 else None,
                "world_size": world,
            },
            "roofline": {
                "kernel": "k_mbest_step1_pair" if win_per_wg == 2 else "k_mbest_step1<double, true>",
                "bound": "lds",
                "achieved": lds_bytes / sec / 1e9 if sec > 0 else 0.0,
                "peak": LDS_PEAK_GBS,
                "unit": "GB/s",
                "frac": lds_bytes / sec / 1e9 / LDS_PEAK_GBS if sec > 0 else 0.0,
                "achieved_definition": "physical bytes the fold passes read from LDS per launch (workgroup sweeps x "
                "passes_per_sweep x N x bytes per LDS element; one pass yields up to 3 periods, and one 8-byte element of "
                "the pair kernel carries a float sample of two windows) / launch time",
                "launch_ms": k1_ms,
                "launch_ms_max_over_ranks": k1_max,
                "step2_launch_ms": k2_ms,
                "lds_bytes_per_launch": lds_bytes,
                "algorithmic_bytes_per_launch": logical,
                "logical_GBs": logical / sec / 1e9 if sec > 0 else 0.0,
                "logical_hbm_ratio": logical / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
                "logical_lds_ratio": logical / sec / 1e9 / LDS_PEAK_GBS if sec > 0 else 0.0,
                "traffic": traffic,
                "traffic_recorded": traffic is not None,
                "traffic_source": tsrc,
                # both numbers of this ratio come from the separate PMC capture (its own launch time)
                "hbm_frac_of_that_capture": (traffic / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and t_ms) else None,
                # step 1 reads the windows and hands the basis rows to step 2 in compact form (the first p <= N/3
                # elements of each row); step 2 writes the (num, N) matrix once (its launch: step2_launch_ms)
                "compulsory_hbm_bytes_per_launch": WINDOWS_PER_GPU * N_SAMPLES * 8 + WINDOWS_PER_GPU * NUM_PERIODS * (N_SAMPLES // 3) * 8,
                "compulsory_hbm_bytes_step2": WINDOWS_PER_GPU * NUM_PERIODS * (N_SAMPLES // 3 + N_SAMPLES) * 8,
                # > 1: the pair kernel's fp64 residuals make one round trip through the HBM workspace per sweep and window
                "traffic_over_compulsory": (traffic / (WINDOWS_PER_GPU * N_SAMPLES * 8 + WINDOWS_PER_GPU * NUM_PERIODS * (N_SAMPLES // 3) * 8)) if traffic else None,
                "note": "SURVEY 8d's logical figure (N*8 B per window-projection against 8 TB/s) is logical_hbm_ratio; it "
                "exceeds 1 because the fused sweep serves every pass but the first from LDS.  The kernel is bound by "
                "instruction issue (VALU + LDS pipe): frac is the physical LDS utilisation, logical_lds_ratio what an "
                "fp64 one-period-per-pass fold would need.",
            },
            "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = line["value"] / cpu["value"]
    # =============================== the fp64 all-norms sweep (north_star's literal kernel) ===============================
    if world == 1:
        warm_clock(lambda: eng.sweep(x, 2, N_SAMPLES // 3, 0))
        eng.profile(True)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            sw_out = eng.sweep(x, 2, N_SAMPLES // 3, 0)
        torch.cuda.synchronize(dev)
        ms_sw = 1e3 * (time.perf_counter() - t0) / reps
        ks_ms, _ = kernel_ms(eng.profile_read(), "k_sweep")
        eng.profile(False)
        checks["sweep_fp64_finite"] = bool(torch.isfinite(sw_out).all().item())
        np_sw, nper_sw = eng.sweep_plan_info(2, N_SAMPLES // 3)
        units_sw = WINDOWS_PER_GPU * nper_sw
        lds_sw = WINDOWS_PER_GPU * np_sw * N_SAMPLES * 8
        trs, tss, _ = load_recorded_traffic("k_sweep")
        sec_sw = ks_ms * 1e-3
        line["sweep_fp64"] = {
            "workload": f"ph_sweep norm mode: every ||P_p x|| (Periods.py:501-510), p=2..{N_SAMPLES // 3}, {WINDOWS_PER_GPU} windows x N={N_SAMPLES} fp64, "
                        "resident in HBM, all values written out",
            "kernel": "k_sweep<double, true>",
            "ms": ms_sw,
            "kernel_ms": ks_ms,
            "window_projections_per_s": units_sw / sec_sw if sec_sw > 0 else 0.0,
            "passes_per_sweep": np_sw,
            "lds_frac": lds_sw / sec_sw / 1e9 / LDS_PEAK_GBS if sec_sw > 0 else 0.0,
            "lds_frac_definition": "physical: passes x N x 8 B read from LDS per window / launch time / 150 TB/s",
            "logical_lds_ratio": units_sw * BYTES_PER_WINDOW_PROJECTION / sec_sw / 1e9 / LDS_PEAK_GBS if sec_sw > 0 else 0.0,
            "logical_hbm_ratio": units_sw * BYTES_PER_WINDOW_PROJECTION / sec_sw / 1e9 / HBM_PEAK_GBS if sec_sw > 0 else 0.0,
            "compulsory_hbm_bytes": WINDOWS_PER_GPU * (N_SAMPLES + nper_sw) * 8,
            "traffic": trs,
            "traffic_source": tss,
        }
        del sw_out
    del x, out, periods, powers, bases

    # =============================== config 3 / config 5 on this one GPU ===============================
    if world == 1 and x3_host is not None:
        x3 = torch.from_numpy(x3_host).to(dev)
        warm_clock(lambda: eng.ramanujan_norms(x3, 2, C3_QHI))
        eng.profile(True)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            nr = eng.ramanujan_norms(x3, 2, C3_QHI)
        torch.cuda.synchronize(dev)
        ms3 = 1e3 * (time.perf_counter() - t0) / reps
        k3_ms, _ = kernel_ms(eng.profile_read(), "k_ramanujan")
        eng.profile(False)
        checks["c3_finite"] = bool(torch.isfinite(nr).all().item())
        units = C3_WINDOWS * (C3_QHI - 1)
        tr3, ts3, _ = load_recorded_traffic("k_ramanujan")
        line["c3_single_gpu"] = {
            "workload": f"config 3: RamanujanPeriods.find_periods, {C3_WINDOWS} windows x N={C3_N} fp64, q=2..{C3_QHI}, resident in HBM",
            "ms": ms3,
            "kernel": "k_ramanujan<double, true>",
            "kernel_ms": k3_ms,
            "window_periods_per_s": units / (ms3 * 1e-3),
            "algorithmic_bytes_per_unit": C3_N * 8,
            "logical_lds_ratio": units * C3_N * 8 / (k3_ms * 1e-3) / 1e9 / LDS_PEAK_GBS if k3_ms else None,
            "logical_hbm_ratio": units * C3_N * 8 / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k3_ms else None,
            "compulsory_hbm_bytes": C3_WINDOWS * (C3_N + C3_QHI + 1) * 8,
            "traffic": tr3,
            "traffic_source": ts3,
        }
        del x3, nr
    if world == 1 and x5_host is not None:
        x5 = torch.from_numpy(x5_host).to(dev)
        warm_clock(lambda: eng.qo_find_periods(x5, C5_NUM, C5_THRESH, C5_MIN, C5_MAX, C5_KCAP))
        eng.profile(True)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            o5 = eng.qo_find_periods(x5, C5_NUM, C5_THRESH, C5_MIN, C5_MAX, C5_KCAP)
        torch.cuda.synchronize(dev)
        ms5 = 1e3 * (time.perf_counter() - t0) / reps
        k5_ms, _ = kernel_ms(eng.profile_read(), "k_qo_find")
        eng.profile(False)
        checks["c5_all_windows_ok"] = int(o5[6].abs().max().item()) == 0
        tr5, ts5, _ = load_recorded_traffic("k_qo_find")
        line["c5_single_gpu"] = {
            "workload": f"config 5, one GPU's batch: QOPeriods.find_periods(num={C5_NUM}, thresh={C5_THRESH}, periods {C5_MIN}..{C5_MAX}), "
            f"{C5_WINDOWS} windows x N={C5_N} fp32, resident in HBM",
            "ms": ms5,
            "kernel": "k_qo_find<float>",
            "kernel_ms": k5_ms,
            "windows_per_s": C5_WINDOWS / (ms5 * 1e-3),
            "dictionary_rows_mean": float(o5[3][:, 1].double().mean().item()),
            "compulsory_hbm_bytes": 2 * C5_WINDOWS * C5_N * 4,
            "traffic": tr5,
            "traffic_source": ts5,
        }
        del x5, o5

    # =============================== config 4 ===============================
    if not args.no_c4:
        total = args.c4_windows
        units = total * (N_SAMPLES // 2 - 1)  # nominal projections of the batch
        if world == 1:
            x4 = torch.from_numpy(x4_host).to(dev)
            del x4_host
            warm_clock(lambda: c4_compute(x4))
            eng.profile(True)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                o4 = c4_compute(x4)
            torch.cuda.synchronize(dev)
            ms4 = 1e3 * (time.perf_counter() - t0) / reps
            k4_ms, _ = kernel_ms(eng.profile_read(), "k_small_to_large")
            eng.profile(False)
            checks["c4_capacity_ok"] = int(o4[3].abs().max().item()) == 0
            tr4, ts4, _ = load_recorded_traffic("k_small_to_large_pair")
            if tr4 is None:
                tr4, ts4, _ = load_recorded_traffic("k_small_to_large")
            line["c4"] = {
                "workload": f"config 4 on one GPU: small_to_large({C4_THRESH}), {total} windows x N={N_SAMPLES}, resident in HBM",
                "n_gpus": 1,
                "ms": ms4,
                "kernel_ms": k4_ms,
                "window_projections_per_s": units / (ms4 * 1e-3),
                "accepted_periods_mean": float(o4[0].double().mean().item()),
                "logical_lds_ratio": units * BYTES_PER_WINDOW_PROJECTION / (k4_ms * 1e-3) / 1e9 / LDS_PEAK_GBS if k4_ms else None,
                "traffic_8192_window_shard": tr4,
                "traffic_source": ts4,
                # windows in + counts / periods / powers out; the rest of the recorded traffic is the staging buffer's
                # write-back cache moving fp64 residuals between LDS and the HBM workspace
                "compulsory_hbm_bytes_8192_window_shard": 8192 * (N_SAMPLES * 8 + 4 + C4_CAP * 12 + 4),
                "traffic_over_compulsory": (tr4 / (8192 * (N_SAMPLES * 8 + 4 + C4_CAP * 12 + 4))) if tr4 else None,
            }
        else:
            # strong scaling: batch on rank 0 -> scatter -> compute -> gather
            from pyperiod_amd.dist import gather_rows, run_sharded_pipelined, scatter_windows

            x_root = torch.from_numpy(x4_host).to(dev) if rank == 0 else None
            del x4_host

            def step4():
                return run_sharded_pipelined(c4_compute, x_root, total, N_SAMPLES, torch.float64, dev, pieces=pieces)

            res = step4()  # warm-up
            barrier()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                res = step4()
            barrier()
            wall = all_max(time.perf_counter() - t0) / reps

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
            kms = all_max(k_ms)
            if rank == 0:
                counts, per, pw, st = res
                checks["c4_shapes"] = bool(counts.shape == (total,) and per.shape == (total, C4_CAP) and pw.shape == (total, C4_CAP))
                checks["c4_capacity_ok"] = bool(int(st.abs().max().item()) == 0)
                checks["c4_pipelined_equals_unpipelined"] = bool(torch.equal(counts, g[0]) and torch.equal(per, g[1]))
                c4_compute(x_root[:1024])
                torch.cuda.synchronize(dev)
                eng.profile(True)
                ts = time.perf_counter()
                for _ in range(3):
                    o1 = c4_compute(x_root)
                torch.cuda.synchronize(dev)
                single_ms = 1e3 * (time.perf_counter() - ts) / 3
                single_k, _ = kernel_ms(eng.profile_read(), "k_small_to_large")
                eng.profile(False)
                checks["c4_sharded_equals_single_gpu"] = bool(torch.equal(o1[0], counts) and torch.equal(o1[1], per) and torch.equal(o1[2], pw))
                try:
                    rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
                except Exception:  # noqa: BLE001
                    rccl = None
                line["c4"] = {
                    "workload": c4_workload(world, total),
                    "n_gpus": world,
                    "scaling": "strong",
                    "ms": 1e3 * wall,
                    "window_projections_per_s": units / wall,
                    "scatter_pieces": args.pieces,
                    "backend": args.backend,
                    "rccl_version": rccl,
                    "phases_unpipelined_ms": {"scatter": 1e3 * (t2 - t1), "compute": 1e3 * (t3 - t2), "gather": 1e3 * (t4 - t3),
                                              "sum": 1e3 * (t4 - t1)},
                    "kernel_ms_slowest_rank": kms,
                    "single_gpu": {"ms": single_ms, "kernel_ms": single_k, "window_projections_per_s": units / (single_ms * 1e-3),
                                   "note": "the same batch on rank 0's GPU alone, input resident, no collective"},
                    "speedup_vs_single_gpu": single_ms / (1e3 * wall),
                    "speedup_compute_only": single_k / kms if kms else None,
                }
                line["strong_scaling"] = {
                    "config": "4: small_to_large(0.05), 65 536 windows from rank 0, RCCL scatter -> compute -> RCCL gather (key c4)",
                    "speedup_vs_single_gpu": line["c4"]["speedup_vs_single_gpu"],
                    "speedup_compute_only": line["c4"]["speedup_compute_only"],
                    "window_projections_per_s": line["c4"]["window_projections_per_s"],
                    "note": "`value` is config 2, weak scaling, no collective -- it grows ~N x by construction and is not the "
                            "scaling result; this block is.  Not comparable with round-2 lines, whose `value` was config 4.",
                }
            barrier()

    failed = [k for k, v in checks.items() if not v]
    if line is not None:
        line["secondary_legs_clock"] = f"sweep_fp64, c3, c5 and c4 at N = 1 run their own call for {WARM_S} s before they are timed (idle clock otherwise)"
        line["checks"] = checks
        print(json.dumps(line), flush=True)
    if world > 1:
        nfail = all_sum(len(failed))  # every rank leaves with the same code
        dist.destroy_process_group()
        if nfail:
            raise SystemExit(f"bench: consistency checks failed: {failed}")
    elif failed:
        raise SystemExit(f"bench: consistency checks failed: {failed}")


if __name__ == "__main__":
    main()
