#!/bin/bash
# tools/ram_pmc.sh <out-dir>: counters of k_ramanujan at config 3 for the shipped kernel and the folds-only variants
# (run from the repo root on the GPU box; rocprofv3 with --pmc only, program directly after `--`).
O="$1"; R="$PWD"; mkdir -p "$O"
( rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS" | sort -u ) > "$O/lds_counters_available.txt"
cd /tmp && export TMPDIR=/tmp
for v in intree RAMFOLDS RAMFOLDSV1; do
  if [ $v = intree ]; then unset PYPERIOD_AMD_LIB; else export PYPERIOD_AMD_LIB=$R/_var/lib_$v.so; fi
  timeout -k 10 100 python3 $R/tools/ab_ram.py 2>&1 | grep "^AB" >> "$O/ab.txt"
  i=0
  for set in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU" \
             "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${v}_$i -- python3 $R/tools/ram_pmc.py > $O/pmc_${v}_$i.log 2>&1 || echo "$v set $i failed" >> "$O/ab.txt"
  done
  python3 $R/tools/pmc_summary.py $O/pmc_${v}_1 $O/pmc_${v}_2 $O/pmc_${v}_3 | grep ramanujan > "$O/pmc_$v.csv"
  rm -rf $O/pmc_${v}_1 $O/pmc_${v}_2 $O/pmc_${v}_3
done
cat "$O/ab.txt"
