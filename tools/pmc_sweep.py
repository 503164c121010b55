#!/usr/bin/env python3
"""Run k_sweep a few times (device-resident input) -- target for rocprofv3 --pmc runs.
usage: pmc_sweep.py [p_lo p_hi [mode]]   (PH_PLAN_MAX_M / PYPERIOD_AMD_LIB from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 1365
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng = PeriodEngine(0)
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).to("cuda:0")
for _ in range(3):
    out = eng.sweep(x, lo, hi, mode)
torch.cuda.synchronize()
print(float(out.sum()))
