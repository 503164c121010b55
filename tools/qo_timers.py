"""In-kernel phase timers of k_qo_find (-DPH_QO_TIMERS build, loaded through PYPERIOD_AMD_LIB): sweep / rhs / CG / reconstruction
times, CG iterations and dictionary sizes of the first twelve windows of a 64-window config-5 batch."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
x = torch.from_numpy(multi_sinusoid_batch(0, 64, 16384, dtype=np.float32)).cuda()
eng = PeriodEngine(0)
eng.qo_find_periods(x, 3, 0.1, 8, 300, 1024)
torch.cuda.synchronize()
