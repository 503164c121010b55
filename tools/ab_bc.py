#!/usr/bin/env python3
"""A/B of best_correlation in one process: one-window kernel (PH_BC_PAIR=0) against the window-pair screen."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).cuda()
res = {}
for name, env in (("single", "0"), ("pair", "1")):
    os.environ["PH_BC_PAIR"] = env
    eng = PeriodEngine(0)
    out = eng.best_correlation(x, 3)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(5):
        out = eng.best_correlation(x, 3)
    torch.cuda.synchronize()
    ks = [ms for nm, ms in eng.profile_read() if nm == "k_best_correlation"]
    eng.profile(False)
    res[name] = [o.cpu().numpy() for o in out]
    print(f"{name:6s} best_correlation(num=3), {W} windows: {np.mean(ks):.3f} ms", flush=True)
    eng.close()
a, b = res["single"], res["pair"]
print("periods / norms / bases / status identical:", [np.array_equal(u, v) for u, v in zip(a, b)])
