"""Effective shader clock and workgroup placement inside the kernels: build a variant with -DPH_CLOCKS
(hipcc ... -DPH_CLOCKS -o _var/lib_clocks.so) and run  PYPERIOD_AMD_LIB=$PWD/_var/lib_clocks.so python3 tools/clocks.py
The kernels print clock64() (shader cycles) against wall_clock64() (100 MHz) and the hardware id of some workgroups."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
eng = PeriodEngine(0)
x = torch.from_numpy(multi_sinusoid_batch(3, 4096, 8192)).cuda()
eng.ramanujan_norms(x, 2, 512); torch.cuda.synchronize()
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 4096)).cuda()
for _ in range(2):
    eng.m_best(x, 10); torch.cuda.synchronize()
    print("----", flush=True)
