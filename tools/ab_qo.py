#!/usr/bin/env python3
"""Same-box A/B of k_qo_find at the config-5 shape (1024 windows x N = 16384 fp32, num = 3, periods 8..300): HIP-event
kernel time and a checksum of the outputs (PYPERIOD_AMD_LIB=... for the variant, see tools/ab_kernels.py)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyperiod_amd import PeriodEngine  # noqa: E402
from pyperiod_amd.synth import multi_sinusoid_batch  # noqa: E402

tag = os.path.basename(os.environ.get("PYPERIOD_AMD_LIB", "in-tree"))
NUM = int(sys.argv[1]) if len(sys.argv) > 1 else 3  # iterations of the greedy loop (config 5: 3)
eng = PeriodEngine(0)
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 16384, dtype=np.float32)).to("cuda:0")
out = eng.qo_find_periods(x, NUM, 0.1, 8, 300, 1024)
eng.profile(True)
for _ in range(5):
    eng.qo_find_periods(x, NUM, 0.1, 8, 300, 1024)
torch.cuda.synchronize()
v = [ms for nm, ms in eng.profile_read() if nm.startswith("k_qo_find")]
eng.profile(False)
h = hashlib.sha256(out[0].cpu().numpy().tobytes() + out[3].cpu().numpy().tobytes()).hexdigest()[:16]
print("AB", tag, "num", NUM, "reported periods mean %.2f, blocks mean %.2f," % (out[3][:, 0].float().mean().item(), out[3][:, 1].float().mean().item()), "k_qo_find %.4f ms (min %.4f)" % (sum(v) / len(v), min(v)), "periods/counts sha", h, flush=True)
