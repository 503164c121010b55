#!/bin/bash
# tools/ab_lib.sh <variant.so> <tool.py> [args]: the tool with the variant library and with the in-tree one, alternating,
# twice each, in one call on one box (devices of the pool differ by up to 10 %: only same-box numbers compare).
V="$1"; shift
for i in 1 2; do
  echo "== variant $(basename $V)"; PYPERIOD_AMD_LIB="$V" timeout -k 10 200 python3 "$@" 2>&1 | grep -v amdgpu.ids
  echo "== in-tree"; timeout -k 10 200 python3 "$@" 2>&1 | grep -v amdgpu.ids
done
