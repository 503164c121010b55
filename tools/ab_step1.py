#!/usr/bin/env python3
"""A/B of m_best step 1 in one process on one box: the one-window fp64 kernel (PH_STEP1_PAIR=0) against the
window-pair screen (default).  Config 2 batch, both variants' outputs compared, kernel times from the library's
HIP events."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
REPS = 10
x = torch.from_numpy(multi_sinusoid_batch(0, W, N)).cuda()
res = {}
for name, env in (("single", "0"), ("pair", "1")):
    os.environ["PH_STEP1_PAIR"] = env
    eng = PeriodEngine(0)
    for gamma in (False, True):
        out = eng.m_best(x, 10, None, 2, gamma, want_sweeps=True)
        torch.cuda.synchronize()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(REPS):
            out = eng.m_best(x, 10, None, 2, gamma, want_sweeps=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / REPS * 1e3
        prof = eng.profile_read()
        eng.profile(False)
        k1 = [ms for nm, ms in prof if nm == "k_mbest_step1"]
        k2 = [ms for nm, ms in prof if nm == "k_mbest_step2"]
        res[(name, gamma)] = [o.cpu().numpy() for o in out]
        print(f"{name:6s} gamma={gamma!s:5s} step1 {np.mean(k1):.3f} ms (min {np.min(k1):.3f})  step2 {np.mean(k2):.3f} ms  wall {wall:.3f} ms  "
              f"sweeps {res[(name, gamma)][4].sum()}", flush=True)
    eng.close()
for gamma in (False, True):
    a, b = res[("single", gamma)], res[("pair", gamma)]
    same_p = np.array_equal(a[0], b[0])
    bad = np.nonzero((a[0] != b[0]).any(axis=1))[0]
    dpow = np.max(np.abs(a[1] - b[1]) / np.maximum(np.abs(a[1]), 1e-300))
    dbas = np.max(np.abs(a[2] - b[2])) / np.max(np.abs(a[2]))
    print(f"gamma={gamma}: periods equal {same_p} (windows differing: {bad[:10].tolist()}), powers rel {dpow:.2e}, bases rel {dbas:.2e}, "
          f"status {np.array_equal(a[3], b[3])}, sweeps equal {np.array_equal(a[4], b[4])}")
    for w in bad[:3]:
        print("  window", w, a[0][w].tolist(), b[0][w].tolist())
sw = res[("pair", False)][4]
print("sweeps per window histogram:", dict(zip(*[a.tolist() for a in np.unique(sw, return_counts=True)])))
pm = np.maximum(sw[0::2], sw[1::2])
print("pairs: max over pairs of sweeps", int(pm.max()), "mean of pair maxima", float(pm.mean()), "mean per window", float(sw.mean()))
