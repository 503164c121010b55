#!/bin/bash
# tools/micro/pair_pass_pmc.sh <out-file>: time and instruction counts per pass of the variants of pair_pass_bench
# (run from the repo root on the GPU box; rocprofv3 with --pmc only, the program directly after `--`).
R="$PWD"; O="$R/$1"; B="$R/tools/micro/pair_pass_bench"
cd /tmp && export TMPDIR=/tmp
for range in "64 2048" "683 2048" "64 683"; do
  echo "== periods $range" >> "$O"
  timeout -k 10 60 $B -1 $range >> "$O" 2>&1
  for v in 0 1; do
    rm -rf /tmp/ppb_$v
    timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d /tmp/ppb_$v -- $B $v $range > /tmp/ppb_$v.log 2>&1
    python3 - "$v" $range >> "$O" <<'PY'
import csv, glob, sys, collections
v, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(f"/tmp/ppb_{v}/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); n = 0
for row in csv.DictReader(open(f[0])):
    tot[row["Counter_Name"]] += float(row["Counter_Value"])
launches = 3
passes = 2048 * 2 * (hi - lo) * launches          # one wavefront each
print(f"variant {v}: per pass: " + "  ".join(f"{k[9:]} {tot[k] / passes:7.1f}" for k in sorted(tot) if k != "SQ_WAVES"))
PY
  done
done
cat "$O"
