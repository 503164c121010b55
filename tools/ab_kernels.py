#!/usr/bin/env python3
"""Same-box A/B of library builds (devices of the pool differ by a few per cent, so two builds are only comparable
inside one gpurun call).  Build a variant next to the in-tree library, e.g.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSOME_KNOB -Iinclude -o _var/lib_x.so pyperiod_amd/csrc/period_hip.hip
    gpurun -- 'for i in 1 2; do python tools/ab_kernels.py; PYPERIOD_AMD_LIB=$PWD/_var/lib_x.so python tools/ab_kernels.py; done'

(`_var/` is git-ignored but travels with the gpurun snapshot).  Prints the HIP-event kernel times (library profiler) of
the four sweep-family kernels at their BASELINE shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyperiod_amd import PeriodEngine  # noqa: E402
from pyperiod_amd.synth import multi_sinusoid_batch  # noqa: E402

x = torch.from_numpy(multi_sinusoid_batch(0, 8192, 4096)).to("cuda:0")
eng = PeriodEngine(0)
tag = os.path.basename(os.environ.get("PYPERIOD_AMD_LIB", "in-tree"))
res = []


def run(name, fn, kern, reps):
    fn()
    eng.profile(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    v = [ms for nm, ms in eng.profile_read() if nm == kern]
    eng.profile(False)
    res.append("%s %.4f" % (name, sum(v) / len(v)))


run("k_mbest_step1", lambda: eng.m_best(x[:1024], 10), "k_mbest_step1", 8)
run("k_mbest_step2", lambda: eng.m_best(x[:1024], 10), "k_mbest_step2", 8)
run("k_small_to_large(8192)", lambda: eng.small_to_large(x, 0.05, None, False, False, cap=64, want_bases=False), "k_small_to_large", 3)
run("k_sweep", lambda: eng.sweep(x[:1024], 2, 1365, 0), "k_sweep", 5)
run("k_best_correlation", lambda: eng.best_correlation(x[:1024], 3, None), "k_best_correlation", 3)
print("AB", tag, "ms:", " | ".join(res))
