#!/usr/bin/env python3
"""Run k_sweep a few times (device-resident input) -- target for rocprofv3 --pmc runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
eng = PeriodEngine(0)
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).to("cuda:0")
for _ in range(4):
    out = eng.sweep(x, 2, 1365, 0)
torch.cuda.synchronize()
print(float(out.sum()))
