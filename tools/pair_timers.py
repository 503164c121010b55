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
