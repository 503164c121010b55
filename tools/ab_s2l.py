#!/usr/bin/env python3
"""A/B of small_to_large in one process: one-window kernel (PH_S2L_PAIR=0) against the window-pair screen; config-4
shard (8192 windows), outputs compared bit for bit, kernel time from the library's HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).cuda()
res = {}
for name, env in (("single", "0"), ("pair", "1")):
    os.environ["PH_S2L_PAIR"] = env
    eng = PeriodEngine(0)
    for thresh, bases in ((0.05, False), (0.1, True)):
        xx = x if not bases else x[:257]
        out = eng.small_to_large(xx, thresh, None, False, False, cap=32, want_bases=bases, nosync=True)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(3):
            out = eng.small_to_large(xx, thresh, None, False, False, cap=32, want_bases=bases, nosync=True)
        torch.cuda.synchronize()
        ks = [ms for nm, ms in eng.profile_read() if nm == "k_small_to_large"]
        eng.profile(False)
        res[(name, thresh)] = [None if o is None else o.cpu().numpy() for o in out]
        print(f"{name:6s} thresh {thresh} bases {bases}: {np.mean(ks):.3f} ms ({xx.shape[0]} windows), accepted mean {res[(name, thresh)][0].mean():.2f}", flush=True)
    eng.close()
for thresh in (0.05, 0.1):
  for other in ("pair",):
    a, b = res[("single", thresh)], res[(other, thresh)]
    ok = [np.array_equal(u, v) for u, v in zip(a[:3] + a[4:], b[:3] + b[4:])]
    if a[3] is not None:  # bases: the rows the kernels wrote
        ok.append(all(np.array_equal(a[3][w, :k], b[3][w, :k]) for w, k in enumerate(a[0])))
    print(other, "thresh", thresh, "counts / periods / powers / status / (bases) identical to the one-window kernel:", ok)
    if not all(ok):
        bad = np.nonzero(a[0] != b[0])[0]
        print("  windows with different counts:", bad[:10], a[0][bad[:5]], b[0][bad[:5]])
