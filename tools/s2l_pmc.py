"""Counter target for k_small_to_large_pair: dispatch 1 = warm-up (64 windows), dispatch 2 = the screen alone (thresh 10:
nothing is ever flagged; 2047 passes per pair), dispatch 3 = config 4's shard (thresh 0.05).  Run under
rocprofv3 --pmc ... -- python3 tools/s2l_pmc.py   (nothing is spawned here); tools/s2l_pmc_table.py prints per dispatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).cuda()
eng = PeriodEngine(0)
for xs, th in ((x[:64], 0.05), (x, 10.0), (x, 0.05)):
    eng.small_to_large(xs, th, None, False, False, cap=32, want_bases=False, nosync=True)
    torch.cuda.synchronize()
eng.close()
