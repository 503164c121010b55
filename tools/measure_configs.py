#!/usr/bin/env python3
"""Time the other BASELINE configs' kernels on one GPU (device-resident inputs, HIP events from
the library).  Not the headline bench; numbers go to DESIGN.md / profiles/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from pyperiod_amd import PeriodEngine, _ffi
from pyperiod_amd.synth import multi_sinusoid_batch

eng = PeriodEngine(0)
dev = torch.device("cuda", 0)
res = {}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    prof = eng.profile_read()
    eng.profile(False)
    kern = {}
    for name, ms in prof:
        kern[name] = kern.get(name, 0.0) + ms / reps
    return out, wall, kern


which = sys.argv[1:] or ["c2host", "sweep", "c4", "c3", "bc", "k1", "c5"]

if "c2host" in which or "sweep" in which or "bc" in which or "k1" in which:
    xh = multi_sinusoid_batch(0, 1024, 4096)
    xd = torch.from_numpy(xh).to(dev)
if "c2host" in which:
    _, wall, kern = timed(lambda: eng.m_best(xh, 10))
    res["c2_host_path_m_best"] = {"wall_ms": wall * 1e3, "kernels_ms": kern, "note": "numpy in -> numpy out incl. PCIe staging of 32 MiB in, 320 MiB bases out"}
if "sweep" in which:
    for mode, nm in ((0, "norm"), (1, "gamma"), (2, "maxabs")):
        _, wall, kern = timed(lambda: eng.sweep(xd, 2, 1365, mode), 5)
        ms = kern["k_sweep"]
        res[f"sweep_{nm}_1024x4096"] = {"ms": ms, "window_proj_per_s": 1024 * 1364 / (ms * 1e-3), "lds_frac": 1024 * 1364 * 32768 / (ms * 1e-3) / 150e12}
if "bc" in which:
    _, wall, kern = timed(lambda: eng.best_correlation(xd, 3), 2)
    res["best_correlation_num3_1024x4096"] = {"kernels_ms": kern, "window_proj_per_s": 1024 * 3 * 1363 / (kern["k_best_correlation"] * 1e-3)}
if "k1" in which:
    pl = [37, 64, 101, 703, 1329, 1365, 2048, 5]
    _, wall, kern = timed(lambda: eng.project_batch(xd, pl), 5)
    ms = kern["k_project_batch"]
    byts = 1024 * 4096 * 8 * (1 + len(pl))
    res["project_batch_1024x4096x8p"] = {"ms": ms, "hbm_GBs": byts / (ms * 1e-3) / 1e9, "frac_of_8TBs": byts / (ms * 1e-3) / 8e12}
if "c4" in which:
    W = 8192
    x4 = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).to(dev)
    out, wall, kern = timed(lambda: eng.small_to_large(x4, 0.05, want_bases=False), 2)
    ms = kern["k_small_to_large"]
    res["c4_small_to_large_8192x4096"] = {"ms": ms, "window_proj_per_s": W * 2047 / (ms * 1e-3), "mean_accepts": float(out[0].double().mean().item()), "lds_frac": W * 2047 * 32768 / (ms * 1e-3) / 150e12}
if "c3" in which:
    W = 1024
    x3 = torch.from_numpy(multi_sinusoid_batch(0, W, 8192)).to(dev)
    _, wall, kern = timed(lambda: eng.ramanujan_norms(x3, 2, 512), 2)
    ms = kern["k_ramanujan"]
    res["c3_ramanujan_1024x8192_q512"] = {"ms": ms, "window_q_per_s": W * 511 / (ms * 1e-3), "logical_GBs": W * 511 * 65536 / (ms * 1e-3) / 1e9}
if "c5" in which:
    W = 512
    x5 = torch.from_numpy(multi_sinusoid_batch(0, W, 16384, dtype=np.float32)).to(dev)
    out, wall, kern = timed(lambda: eng.qo_find_periods(x5, 3, 0.1, 8, 300, 1024), 2)
    ms = kern["k_qo_find"]
    res["c5_qo_find_periods_512x16384_fp32"] = {"ms": ms, "windows_per_s": W / (ms * 1e-3), "mean_blocks": float(out[3][:, 1].double().mean().item()),
                                                "mean_rows": float(out[2].double().sum(1).mean().item())}
print(json.dumps(res, indent=1))
