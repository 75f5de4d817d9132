#!/bin/bash
# Same-box A/B of environment switches of ONE build on BASELINE config 2 (full time_loop, wall seconds), alternating.
# Usage (GPU box): VARIANTS="LUDVM_SYM_MIXED=0 LUDVM_X=default" REPS=3 bash tools/ab_cfg2_env.sh
for rep in $(seq 1 ${REPS:-3}); do
  for v in $VARIANTS; do
    env $v python tools/run_configs.py cfg2 --no-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['wall_s'],3), d['final_wake'], d['Cl_last'])"
  done
done
