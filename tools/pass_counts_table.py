#!/usr/bin/env python3
"""pass_counts_table.py <cases.json> <pmc_dir> [<trace_dir>] -> markdown table: wave-instructions per pass per window."""
import collections, csv, glob, json, sys
cases = json.load(open(sys.argv[1]))
W = cases["windows"]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "k_sweep"
disp = collections.OrderedDict()
for f in glob.glob(f"{sys.argv[2]}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = []
if len(sys.argv) > 3:
    for f in glob.glob(f"{sys.argv[3]}/**/*_kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
ids = sorted(disp)
assert len(ids) == len(cases["cases"]), (len(ids), len(cases["cases"]))
print("| case | passes | us | LDS frac (passes x N x 8 B / 150 TB/s) | VALU/pass | SALU/pass | LDS/pass | VALU busy | cycles/pass/SIMD |")
print("|---|---|---|---|---|---|---|---|---|")
for k, (i, c) in enumerate(zip(ids, cases["cases"])):
    d = disp[i]; n = W * c["passes"]
    us = dur[k] if dur else float("nan")
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8  # summed over the 8 XCDs
    print(f"| {c['label']} | {c['passes']} | {us:.1f} | {n * 32768 / (us * 1e-6) / 150e12 if dur else 0:.3f} | {d['SQ_INSTS_VALU'] / n:.1f} | "
          f"{d['SQ_INSTS_SALU'] / n:.1f} | {d['SQ_INSTS_LDS'] / n:.1f} | {d['SQ_INSTS_VALU'] * 4 / 1024 / cyc if cyc else 0:.2f} | {cyc * 1024 / n if cyc else 0:.0f} |")
