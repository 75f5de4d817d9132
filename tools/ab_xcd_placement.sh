#!/bin/bash
# Same-box A/B of two builds (LIBS="old.so new.so"): roll-up sweep over wake sizes whose tile count is and is not a multiple of 8,
# the headline call, config 2.  Usage (GPU box): LIBS="_ab/prev/libludvm_hip.so ludvm_amd/csrc/libludvm_hip.so" bash tools/ab_xcd_placement.sh
SIZES="${SIZES:-20000 30000 40960 41472 43008 50000 57344 60000 65536 66048 66560 67584 70000 90000 100000 131072 150000 200000 300000 500000}"
for rep in 1 2; do for lib in $LIBS; do echo "== $lib sweep $rep"
  LUDVM_HIP_LIB=$lib SWEEP_SECONDS=0.2 SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"
done; done
for rep in 1 2 3; do for lib in $LIBS; do echo "== $lib bench $rep"
  LUDVM_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --cpu-rows 0 --repeats 1 --cfg4-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
for rep in 1 2; do for lib in $LIBS; do echo "== $lib config 2, $rep"
  LUDVM_HIP_LIB=$lib python tools/run_configs.py cfg2 --no-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],3), d['final_wake'], d['Cl_last'])"
done; done
