"""Workgroup start / end stamps of k_small_to_large_pair (-DPH_CLOCKS build, loaded through PYPERIOD_AMD_LIB): the
kernel stores {start, end (100 MHz), hardware id, accepts} per workgroup in a device array, read back here.
usage: s2l_clocks.py [windows] -> summary: makespan, duration spread, resident workgroups over time, the tail."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from pyperiod_amd import PeriodEngine, _ffi
from pyperiod_amd.synth import multi_sinusoid_batch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.from_numpy(multi_sinusoid_batch(0, min(W, 8192), 4096)).cuda()
if W > 8192:
    x = x.repeat(W // 8192, 1)
eng = PeriodEngine(0)
for _ in range(2):
    eng.small_to_large(x, 0.05, None, False, False, cap=32, want_bases=False, nosync=True)
torch.cuda.synchronize()
nwg = (W + 1) // 2
a = np.zeros((nwg, 4), dtype=np.int64)
lib = _ffi.load()
rc = lib.ph_debug_stamps(C.c_void_p(a.ctypes.data), C.c_int(nwg))
assert rc == 0
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0  # microseconds
dur = en - st
print(f"windows {W}: workgroups {nwg}  makespan {en.max():.0f} us  sum of durations {dur.sum() / 1e3:.1f} ms  mean {dur.mean():.0f} us  "
      f"min {dur.min():.0f}  p10 {np.percentile(dur, 10):.0f}  median {np.median(dur):.0f}  p90 {np.percentile(dur, 90):.0f}  max {dur.max():.0f}")
ev = sorted([(t, 1) for t in st] + [(t, -1) for t in en])
cur, peak, prof = 0, 0, []
for t, d in ev:
    cur += d
    peak = max(peak, cur)
    prof.append((t, cur))
T = en.max()
print(f"peak resident workgroups {peak};  sum / peak = {dur.sum() / peak:.0f} us = makespan with no tail;  last start {st.max():.0f} us")
ts = np.array([p[0] for p in prof])
for frac in (0.25, 0.5, 0.7, 0.8, 0.85, 0.9, 0.95, 0.98):
    k = np.searchsorted(ts, frac * T, side="right") - 1
    print(f"  at {frac:.2f} of the makespan ({frac * T:.0f} us): {prof[k][1]} workgroups resident")
order = np.argsort(st)
for i, chunk in enumerate(np.array_split(order, 8)):
    print(f"  start octile {i}: start {st[chunk].mean():.0f} us  duration mean {dur[chunk].mean():.0f}  max {dur[chunk].max():.0f}")
acc = a[:, 3]
for lo, hi in ((0, 20), (20, 26), (26, 32), (32, 40), (40, 99)):
    m = (acc >= lo) & (acc < hi)
    if m.any():
        print(f"  accepts of the pair in [{lo}, {hi}): {m.sum()} workgroups, duration mean {dur[m].mean():.0f} us")
