#!/bin/bash
# usage: run.sh "<lib>:<block>" ...
for spec in "$@"; do
  lib=${spec%%:*}; blk=${spec#*:}
  PYPERIOD_AMD_LIB=$PWD/build_variants/lib_$lib.so PH_SWEEP_BLOCK=$blk timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/v_$lib_$blk.log 2>&1
  echo "$spec $(grep -o '"launch_ms": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/v_$lib_$blk.log | tr '\n' ' ')"
done
