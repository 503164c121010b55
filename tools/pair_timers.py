"""In-kernel phase timers of k_mbest_step1_pair: build a variant with -DPH_PAIR_TIMERS (hipcc ... -o _var/lib_timers.so) and run
    PYPERIOD_AMD_LIB=$PWD/_var/lib_timers.so python3 tools/pair_timers.py
The first six workgroups print screen / scan / load / exact / update times (100 MHz ticks), candidates and sweeps."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).cuda()
eng = PeriodEngine(0)
for gamma in (False, True):
    print("gamma", gamma, flush=True)
    eng.m_best(x, 10, None, 2, gamma)
    torch.cuda.synchronize()
