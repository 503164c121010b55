"""In-kernel phase timers of k_mbest_step2: build a variant with -DPH_STEP2_TIMERS (hipcc ... -o _var/lib_s2t.so) and run
    PYPERIOD_AMD_LIB=$PWD/_var/lib_s2t.so python3 tools/step2_timers.py
Every 128th workgroup prints staging / fold / flush+scan / split times (100 MHz ticks), rows examined and splits."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
eng = PeriodEngine(0)
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).cuda()
eng.m_best(x, 10); torch.cuda.synchronize()
print("----")
eng.m_best(x, 10); torch.cuda.synchronize()
