#!/bin/bash
# tools/micro/pair_pass_pmc.sh <out-file>: time and instruction counts per pass of the variants of pair_pass_bench
# (run from the repo root on the GPU box; rocprofv3 with --pmc only, the program directly after `--`).
R="$PWD"; O="$R/$1"; B="$R/tools/micro/pair_pass_bench"
cd /tmp && export TMPDIR=/tmp
for job in "64 2048 -1" "683 2048 -1" "64 683 -1" "342 683 2" "65 342 3"; do
  set -- $job
  echo "== periods [$1, $2)" >> "$O"
  timeout -k 10 60 $B $3 $1 $2 >> "$O" 2>&1
  rm -rf /tmp/ppb
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/ppb -- $B $3 $1 $2 > /tmp/ppb.log 2>&1
  python3 - $1 $2 >> "$O" <<'PY'
import csv, glob, sys, collections, re
lo, hi = int(sys.argv[1]), int(sys.argv[2])
f = glob.glob("/tmp/ppb/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f[0])):
    m = re.search(r"k<(\d+)>", row["Kernel_Name"])
    tot[m.group(1) if m else row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
passes = 2048 * 2 * (hi - lo) * 3  # 3 launches, one wavefront per pass
for v in sorted(tot):
    print(f"variant {v}: per pass: " + "  ".join(f"{k[9:]} {tot[v][k] / passes:7.1f}" for k in sorted(tot[v])))
PY
done
cat "$O"
