"""Counter target for k_ramanujan at config 3 (4096 windows x N = 8192, q <= 512): warm-up + 2 launches.  Run under
rocprofv3 --pmc ... -- python3 tools/ram_pmc.py (nothing is spawned here).  PYPERIOD_AMD_LIB selects a variant
(-DPH_RAM_FOLDS_ONLY: the root folds alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
x = torch.from_numpy(multi_sinusoid_batch(3, 4096, 8192)).cuda()
eng = PeriodEngine(0)
for _ in range(3):
    eng.ramanujan_norms(x, 2, 512)
    torch.cuda.synchronize()
eng.close()
