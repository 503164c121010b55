#!/bin/bash
# tools/pass_counts_pair.sh <out-subdir-of-gpurun_out>: per-pass-type instruction counts of k_mbest_step1_pair
[ -n "$1" ] || { echo "usage: tools/pass_counts_pair.sh <out-subdir-of-gpurun_out>"; exit 2; }
R="$PWD"; O="$R/gpurun_out/$1"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/tools/pass_counts_pair.py" "$O/cases.json" > "$O/trace.log" 2>&1 || { tail -5 "$O/trace.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc" -- python3 "$R/tools/pass_counts_pair.py" "$O/cases.json" > "$O/pmc.log" 2>&1 || { tail -5 "$O/pmc.log"; exit 1; }
cd "$R"
python3 tools/pass_counts_table.py "$O/cases.json" "$O/pmc" "$O/trace" k_mbest_step1_pair > "$O/table.md" && cat "$O/table.md"
rm -rf "$O/trace" "$O/pmc"
