"""Dictionary sizes of the config-5 batch (1024 windows x N = 16384 fp32, num = 3, periods 8..300): how many windows stop after
their first period, how many rows K the others solve for, the largest ones."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch
x = torch.from_numpy(multi_sinusoid_batch(0, 1024, 16384, dtype=np.float32)).cuda()
eng = PeriodEngine(0)
out = eng.qo_find_periods(x, 3, 0.1, 8, 300, 1024)
per = out[0].cpu().numpy(); keeps = out[2].cpu().numpy(); cnt = out[3].cpu().numpy(); st = out[6].cpu().numpy()
K = keeps.sum(1)
print("status counts", np.unique(st, return_counts=True))
print("blocks", np.unique(cnt[:,1], return_counts=True))
print("K quantiles", np.percentile(K, [0, 25, 50, 75, 90, 99, 100]))
h = K[cnt[:,1]==3]
print("K of 3-block windows: n", len(h), "quantiles", np.percentile(h, [0, 25, 50, 75, 90, 99, 100]))
big = np.argsort(-K)[:8]
print("largest", [(int(i), int(K[i]), per[i].tolist()) for i in big])
