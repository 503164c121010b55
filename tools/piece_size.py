#!/usr/bin/env python3
"""How small may the pieces of a rank's block be?  run_sharded_pipelined (pyperiod_amd/dist.py) cuts the 8192 windows a
rank owns at config 4 / 8 GPUs into pieces so that the scatter of piece k+1 overlaps the kernel on piece k; every piece
is one launch of k_small_to_large, and a 1024-window piece is exactly one wave of workgroups on the chip.  This prints
the kernel time (HIP events of the library) of one shard as 1, 2, 4, 8 and 16 launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyperiod_amd import PeriodEngine  # noqa: E402
from pyperiod_amd.synth import multi_sinusoid_batch  # noqa: E402

W = 8192
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).to("cuda:0")
eng = PeriodEngine(0)


def run(pieces):
    step = W // pieces
    for k in range(pieces):
        eng.small_to_large(x[k * step : (k + 1) * step], 0.05, None, False, False, cap=32, want_bases=False, nosync=True)


for pieces in (1, 2, 4, 8, 16):
    run(pieces)
    torch.cuda.synchronize()
    eng.profile(True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    ev0.record()
    for _ in range(reps):
        run(pieces)
    ev1.record()
    torch.cuda.synchronize()
    ks = [ms for nm, ms in eng.profile_read() if nm == "k_small_to_large"]
    eng.profile(False)
    print(f"pieces {pieces:2d} x {W // pieces:5d} windows: kernels {sum(ks) / reps:7.3f} ms per shard, stream time {ev0.elapsed_time(ev1) / reps:7.3f} ms", flush=True)
