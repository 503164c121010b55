#!/usr/bin/env python3
"""One k_sweep launch per pass type (1024 windows x N=4096 fp64), meant to run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ...`: the per-dispatch counters divided by
(windows x passes) give wave-instructions per pass for every pass type.  `tools/pass_counts_table.py`
turns the counter CSV into the table.  Tuning aid, not a bench."""
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
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rows = []
for label, lo, hi, mm in CASES:
    os.environ["PH_PLAN_MAX_M"] = str(mm)
    eng = PeriodEngine(0)
    n_pass, n_per = eng.sweep_plan_info(lo, hi)
    eng.sweep(x, lo, hi, mode)
    torch.cuda.synchronize()
    rows.append({"label": label, "p_lo": lo, "p_hi": hi, "max_m": mm, "periods": hi - lo + 1, "passes": n_pass})
    eng.close()
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "pass_counts_cases.json")
json.dump({"windows": W, "n": 4096, "cases": rows}, open(out, "w"), indent=1)
print("launched", len(rows), "cases")
