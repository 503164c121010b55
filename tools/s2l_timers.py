"""In-kernel phase timers of k_small_to_large and k_small_to_large_pair (-DPH_S2L_TIMERS build, loaded through PYPERIOD_AMD_LIB):
screen / exact / update times, rounds, events and accepts of the first workgroups.  usage: s2l_timers.py [windows]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).cuda()
for env in ("0", "1"):
    os.environ["PH_S2L_PAIR"] = env
    eng = PeriodEngine(0)
    print("PH_S2L_PAIR", env, flush=True)
    eng.small_to_large(x, 0.05, None, False, False, cap=32, want_bases=False, nosync=True)
    torch.cuda.synchronize()
    eng.close()
