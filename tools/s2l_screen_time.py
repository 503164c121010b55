"""Screen-alone and config-4 shard time of k_small_to_large_pair (HIP events), for A/B runs of two builds on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
x = torch.from_numpy(multi_sinusoid_batch(0, 8192, 4096)).cuda()
eng = PeriodEngine(0)
for name, th in (("screen alone (thresh 10)", 10.0), ("shard (thresh 0.05)", 0.05)):
    eng.small_to_large(x, th, None, False, False, cap=32, want_bases=False, nosync=True)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(5):
        eng.small_to_large(x, th, None, False, False, cap=32, want_bases=False, nosync=True)
    torch.cuda.synchronize()
    v = [ms for nm, ms in eng.profile_read() if nm == "k_small_to_large"]
    eng.profile(False)
    print(os.path.basename(os.environ.get("PYPERIOD_AMD_LIB", "in-tree")), name, "%.3f ms (min %.3f)" % (sum(v) / len(v), min(v)), flush=True)
