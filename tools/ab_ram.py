#!/usr/bin/env python3
"""Same-box A/B of k_ramanujan at the config-3 shape (4096 windows x N = 8192, q <= 512) and at the default range of a
4096-sample window (q <= 1365): HIP-event kernel time and a checksum of the norms, so that two builds
(PYPERIOD_AMD_LIB=... for the variant, see tools/ab_kernels.py) can be compared for speed and for bit equality."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyperiod_amd import PeriodEngine  # noqa: E402
from pyperiod_amd.synth import multi_sinusoid_batch  # noqa: E402

tag = os.path.basename(os.environ.get("PYPERIOD_AMD_LIB", "in-tree"))
eng = PeriodEngine(0)
for name, W, N, q_hi, reps in (("c3 4096x8192 q<=512", 4096, 8192, 512, 5), ("1024x4096 q<=1365", 1024, 4096, 1365, 5)):
    x = torch.from_numpy(multi_sinusoid_batch(3, W, N)).to("cuda:0")
    out = eng.ramanujan_norms(x, 2, q_hi)
    eng.profile(True)
    for _ in range(reps):
        eng.ramanujan_norms(x, 2, q_hi)
    torch.cuda.synchronize()
    v = [ms for nm, ms in eng.profile_read() if nm == "k_ramanujan"]
    eng.profile(False)
    h = hashlib.sha256(out[:, 2:].cpu().numpy().tobytes()).hexdigest()[:16]
    print("AB", tag, name, "k_ramanujan %.4f ms (min %.4f)" % (sum(v) / len(v), min(v)), "norms sha", h, flush=True)
