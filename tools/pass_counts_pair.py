#!/usr/bin/env python3
"""Per-pass-type instruction counts of the window-pair screen: one m_best(num=1) launch (one sweep per window, plus the
small exact phase) per period range / plan type, meant to run under `rocprofv3 --pmc ...` and `--kernel-trace`;
`tools/pass_counts_table.py <cases> <pmc> <trace> k_mbest_step1_pair` prints the table.  Tuning aid, not a bench."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyperiod_amd import PeriodEngine
from pyperiod_amd.synth import multi_sinusoid_batch

W = 1024
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).to("cuda:0")
CASES = [  # (label, p_lo, p_hi, max_m)
    ("single R3 p1366-2047", 1366, 2047, 1),
    ("single R4-6 p683-1365", 683, 1365, 1),
    ("single R7-12 p342-682", 342, 682, 1),
    ("single R13-24 p171-341", 171, 341, 1),
    ("single R25-64 p64-170", 64, 170, 1),
    ("single p2-63", 2, 63, 1),
    ("plan m2 p342-1365", 342, 1365, 2),
    ("plan m4 p171-1365", 171, 1365, 4),
    ("plan m4 p2-1365 (full)", 2, 1365, 4),
    ("plan m1 p2-1365", 2, 1365, 1),
]
rows = []
for label, lo, hi, mm in CASES:
    os.environ["PH_PLAN_MAX_M"] = str(mm)
    eng = PeriodEngine(0)
    n_pass, n_per = eng.sweep_plan_info(lo, hi)
    eng.m_best(x, 1, hi, lo)
    torch.cuda.synchronize()
    rows.append({"label": label, "p_lo": lo, "p_hi": hi, "max_m": mm, "periods": hi - lo + 1, "passes": n_pass})
    eng.close()
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pass_counts_pair_cases.json")
json.dump({"windows": W // 2, "n": 4096, "cases": rows}, open(out, "w"), indent=1)  # a wave-pass serves a PAIR of windows
print("launched", len(rows), "cases")
