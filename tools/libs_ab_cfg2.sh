#!/bin/bash
# Same-box A/B of two builds of the library on config 2 (full time_loop) -- alternating, three rounds:
#   bash tools/libs_ab_cfg2.sh <libA.so> <libB.so>      (python-side switch LUDVM_HIP_LIB; run on the GPU box)
set -o pipefail
A=$1; B=$2
for r in 1 2 3; do
  for L in $A $B; do
    LUDVM_HIP_LIB=$L python3 tools/run_configs.py cfg2 --no-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['time_loop_s'],3), d['Cl_last'])"
  done
done
