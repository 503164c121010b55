#!/bin/bash
# tools/micro/pair_pass_pmc.sh <out-file>: time and instruction counts per pass of pair_pass_bench
# (run from the repo root on the GPU box; rocprofv3 with --pmc only, the program directly after `--`).
R="$PWD"; O="$R/$1"; B="$R/tools/micro/pair_pass_bench"
cd /tmp && export TMPDIR=/tmp
for job in "1 683 2048" "1 64 683" "2 342 683" "4 65 342"; do
  set -- $job
  timeout -k 10 60 $B $1 $2 $3 >> "$O" 2>&1
  rm -rf /tmp/ppb
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/ppb -- $B $1 $2 $3 > /tmp/ppb.log 2>&1
  python3 - $2 $3 >> "$O" <<'PY'
import csv, glob, sys, collections
lo, hi = int(sys.argv[1]), int(sys.argv[2])
f = glob.glob("/tmp/ppb/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float)
for row in csv.DictReader(open(f[0])):
    if "k<" in row["Kernel_Name"]:
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
passes = 2048 * 2 * (hi - lo) * 3  # 3 launches, one wavefront per pass
print("    instructions per pass: " + "  ".join(f"{k[9:]} {tot[k] / passes:6.1f}" for k in sorted(tot)))
PY
done
cat "$O"
