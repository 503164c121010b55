#!/usr/bin/env python3
"""Per-call latency of the drop-in class surface for ONE window (numpy in, numpy out): what a
reference user who switches the import sees.  Not a throughput bench."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pyperiod_amd import Periods, QOPeriods, RamanujanPeriods
from pyperiod_amd.synth import multi_sinusoid_window

x = multi_sinusoid_window(1, 4096)
p = Periods()


def lat(name, fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / reps:8.3f} ms per call")


lat("Periods.project(x, 37)", lambda: Periods.project(x, 37))
lat("Periods.project(x, 36, orth)", lambda: Periods.project(x, 36, False, True))
lat("Periods.periodic_norm(x)", lambda: Periods.periodic_norm(x))
lat("Periods().m_best(x, 10)", lambda: p.m_best(x, num=10))
lat("Periods().m_best_gamma(x, 10)", lambda: p.m_best_gamma(x, num=10))
lat("Periods().small_to_large(x, .05)", lambda: p.small_to_large(x, thresh=0.05))
lat("Periods().best_correlation(x, 3)", lambda: p.best_correlation(x, num=3), 5)
lat("Periods().best_frequency(x, num=5)", lambda: p.best_frequency(x, num=5), 5)
lat("RamanujanPeriods.find_periods 2..512", lambda: RamanujanPeriods().find_periods(x, 2, 512), 5)
lat("QOPeriods.find_periods(num=3)", lambda: QOPeriods().find_periods(x, num=3, thresh=0.05, max_length=300), 5)
