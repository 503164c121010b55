#!/usr/bin/env python3
"""Run every kernel of libperiod_hip.so at its BASELINE config shape (device-resident inputs) --
the target program of the rocprofv3 passes in tools/profile_all.sh, and (without a profiler)
the source of the per-kernel HIP-event timings.

usage: profile_all.py [--reps R] [--json OUT] [kernel-group ...]
groups: mbest sweep s2l bc ram k1 qo bf orth fold misc   (default: all)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch

GROUPS = ["mbest", "sweep", "s2l", "bc", "ram", "k1", "qo", "bf", "orth", "fold", "misc"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--json", default=None)
    ap.add_argument("groups", nargs="*")
    a = ap.parse_args()
    which = a.groups or GROUPS
    eng = PeriodEngine(0)
    dev = torch.device("cuda", 0)
    res = {}

    def run(tag, fn, units=None, unit_name=None, alg_bytes=None):
        fn()  # warm-up (tables, workspaces)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        kern = {}
        for name, ms in prof:
            kern.setdefault(name, []).append(ms)
        rec = {"kernels_ms": {k: sum(v) / a.reps for k, v in kern.items()}, "launches_per_call": {k: len(v) // a.reps for k, v in kern.items()}}
        tot = sum(rec["kernels_ms"].values())
        if units:
            rec[unit_name + "_per_s"] = units / (tot * 1e-3)
        if alg_bytes:
            rec["logical_GBs"] = alg_bytes / (tot * 1e-3) / 1e9
            rec["logical_over_lds_peak"] = rec["logical_GBs"] / 150000.0
        res[tag] = rec
        print(tag, json.dumps(rec), flush=True)

    x4k = None
    if set(which) & {"mbest", "sweep", "bc", "k1", "bf", "orth", "misc"}:
        x4k = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).to(dev)
    if "mbest" in which:  # config 2
        out = eng.m_best(x4k, 10, want_sweeps=True)
        sw = int(out[4].sum().item())
        run("c2_m_best_1024x4096", lambda: eng.m_best(x4k, 10), sw * 1364, "window_proj", sw * 1364 * 32768)
        run("c2_m_best_gamma_1024x4096", lambda: eng.m_best(x4k, 10, gamma=True))
    if "sweep" in which:
        for mode, nm in ((0, "norm"), (2, "maxabs")):
            run(f"sweep_{nm}_1024x4096", lambda: eng.sweep(x4k, 2, 1365, mode), 1024 * 1364, "window_proj", 1024 * 1364 * 32768)
    if "s2l" in which:  # config 4, one GPU's shard
        x = torch.from_numpy(multi_sinusoid_batch(0, 8192, 4096)).to(dev)
        run("c4_small_to_large_8192x4096", lambda: eng.small_to_large(x, 0.05, cap=32, want_bases=False, nosync=True), 8192 * 2047, "window_proj", 8192 * 2047 * 32768)
        del x
    if "bc" in which:
        run("best_correlation_num3_1024x4096", lambda: eng.best_correlation(x4k, 3), 1024 * 3 * 1363, "window_proj", 1024 * 3 * 1363 * 32768)
    if "ram" in which:  # config 3 at its stated batch
        x = torch.from_numpy(multi_sinusoid_batch(0, 4096, 8192)).to(dev)
        run("c3_ramanujan_4096x8192_q512", lambda: eng.ramanujan_norms(x, 2, 512), 4096 * 511, "window_q", 4096 * 511 * 65536)
        del x
    if "k1" in which:
        for pl in ([37, 64, 101, 703, 1329, 1365, 2048, 5], list(range(30, 94))):
            byts = 1024 * 4096 * 8 * (1 + len(pl))
            run(f"project_batch_1024x4096x{len(pl)}p", lambda: eng.project_batch(x4k, pl), 1024 * len(pl), "window_proj", None)
            res[f"project_batch_1024x4096x{len(pl)}p"]["hbm_GBs_compulsory"] = byts / (res[f"project_batch_1024x4096x{len(pl)}p"]["kernels_ms"]["k_project_batch"] * 1e-3) / 1e9
    if "qo" in which:  # config 5: fp32 N=16384 (the stated 1024 windows per GPU)
        x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 16384, dtype=np.float32)).to(dev)
        run("c5_qo_find_1024x16384_fp32", lambda: eng.qo_find_periods(x, 3, 0.1, 8, 300, 1024), 1024, "windows")
        del x
    if "fold" in which:
        x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 16384, dtype=np.float32)).to(dev)
        run("c5_fold_sums_1024x16384_fp32", lambda: eng.fold_sums(x, [37, 64, 101], [37, 63, 100]), 1024 * 3, "window_proj")
        wts = torch.randn(1024, 200, dtype=torch.float64, device=dev)
        run("c5_tile_sum_1024x16384_fp32", lambda: eng.tile_sum(wts, 16384, [37, 64, 101], [37, 63, 100], np.float32), 1024, "windows")
        del x
    if "bf" in which:
        run("best_frequency_num5_256x4096", lambda: eng.best_frequency(x4k[:256], None, 5), 256, "windows")
        x2k = torch.from_numpy(multi_sinusoid_batch(0, 256, 2000)).to(dev)  # the README length: not a power of two
        run("best_frequency_num5_256x2000", lambda: eng.best_frequency(x2k, None, 5), 256, "windows")
    if "orth" in which:
        run("orth_powers_256x4096", lambda: eng.orth_powers(x4k[:256]), 256, "windows")
    if "misc" in which:
        run("periodic_norm_1024x4096", lambda: eng.periodic_norm(x4k), 1024, "windows")
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
