#!/usr/bin/env python3
"""Time k_sweep on period sub-ranges under different pass plans: cost per pass type
(device-resident input, HIP events from the library).  Tuning aid, not a bench."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch

x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).to("cuda:0")
cases = [  # (label, p_lo, p_hi, max_m, passes)
    ("single R4-6", 683, 1365, 1, 683),
    ("single R7-12", 342, 682, 1, 341),
    ("single R13-24", 171, 341, 1, 171),
    ("single R25-64", 64, 170, 1, 107),
    ("small p<64", 2, 63, 1, 62),
    ("m2 base 342-682", 342, 1365, 2, None),
    ("m4 base 171-341", 171, 1365, 4, None),
    ("full", 2, 1365, 4, None),
]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for label, lo, hi, mm, _ in cases:
    os.environ["PH_PLAN_MAX_M"] = str(mm)
    eng = PeriodEngine(0)
    eng.sweep(x, lo, hi, mode)
    eng.profile(True)
    for _ in range(5):
        eng.sweep(x, lo, hi, mode)
    ms = sum(t for _, t in eng.profile_read()) / 5
    eng.profile(False)
    print(f"{label:18s} p=[{lo},{hi}] max_m={mm}: {ms*1e3:8.1f} us  ({ms*1e6/1024/(hi-lo+1):6.2f} ns per window-period)")
    eng.close()
