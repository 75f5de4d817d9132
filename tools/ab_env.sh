#!/bin/bash
# Same-box A/B of environment switches of ONE build: roll-up sweep (symmetric fp32 / hi+lo columns) under each setting,
# alternating.  Usage (GPU box): VARIANTS="LUDVM_SYM_MIXED=1 LUDVM_SYM_MIXED=0" bash tools/ab_env.sh
SIZES="${SIZES:-24576 32768 40960 49152 65536 98304 131072 262144}"
for rep in 1 2 3; do
  for v in $VARIANTS; do
    echo "== $v sweep $rep"
    env $v python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], d['sym_f32_us'], d['sym_f32x2_us'])"
  done
done
for rep in 1 2; do
  for v in $VARIANTS; do
    echo "== $v bench $rep"
    env $v python bench.py --steps 10 --warmup 2 --cpu-rows 0 --repeats 1 --cfg4-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
  done
done
