#!/bin/bash
# Same-box A/B of builds of libludvm_hip.so (LUDVM_HIP_LIB) on BASELINE config 2 (full time_loop, wall seconds), alternating.
# Usage (GPU box): LIBS="a.so b.so" REPS=2 bash tools/ab_cfg2.sh
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do
    LUDVM_HIP_LIB=$lib python tools/run_configs.py cfg2 --no-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['wall_s'],3), d['final_wake'])"
  done
done
