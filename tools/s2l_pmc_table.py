#!/usr/bin/env python3
"""Per-dispatch counters of k_small_to_large_pair from rocprofv3 --pmc output directories (tools/s2l_pmc.py)."""
import collections, csv, glob, sys
rows = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_small_to_large" in r["Kernel_Name"]:
                key = int(r["Dispatch_Id"])
                rows[(d, key)][r["Counter_Name"]] = rows[(d, key)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
bydisp = collections.defaultdict(dict)
for d in sys.argv[1:]:
    keys = sorted(k for (dd, k) in rows if dd == d)
    for i, k in enumerate(keys):
        bydisp[i].update(rows[(d, k)])
names = {0: "warm-up 64 windows", 1: "screen alone (thresh 10)", 2: "shard thresh 0.05"}
for i in sorted(bydisp):
    c = bydisp[i]
    print(f"dispatch {i} ({names.get(i, '?')}):")
    for k in sorted(c):
        print(f"   {k:28s} {c[k]:.6g}")
    if "SQ_INSTS_LDS" in c and c["SQ_INSTS_LDS"]:
        print(f"   VALU / LDS {c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_LDS']:.2f}   SALU / LDS {c.get('SQ_INSTS_SALU', 0) / c['SQ_INSTS_LDS']:.2f}")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        print(f"   HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {(2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024 / 1e6:.1f} MB")
