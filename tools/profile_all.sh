#!/bin/bash
# Capture rocprofv3 evidence for every kernel (run from the repo root on the GPU box):
#   tools/profile_all.sh <out-subdir-of-gpurun_out> [groups...]
# -> hip-event timings, kernel_stats.csv (--kernel-trace --stats) and pmc_summary.csv
#    (FETCH_SIZE / WRITE_SIZE / SQ counters, separate passes, program directly after `--`).
[ -n "$1" ] || { echo "usage: tools/profile_all.sh <out-subdir-of-gpurun_out> [groups...]"; exit 2; }
R="$PWD"
O="$R/gpurun_out/$1"; shift
G="$@"
rm -rf "$O"; mkdir -p "$O"
python3 tools/profile_all.py --reps 3 --json $O/hip_events.json $G > $O/hip_events.log 2>&1 || { tail -5 $O/hip_events.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/profile_all.py --reps 3 $G > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/tools/profile_all.py --reps 2 $G > $O/pmc_$i.log 2>&1 || { echo "pmc set $i failed"; tail -3 $O/pmc_$i.log; }
done
cd $R
python3 tools/pmc_summary.py $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 > $O/pmc_summary.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4
grep -c . $O/pmc_summary.csv; cut -c1-120 $O/kernel_stats.csv | head -30
