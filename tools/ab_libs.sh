#!/bin/bash
# Same-box A/B of several builds of libludvm_hip.so (LUDVM_HIP_LIB): roll-up sweep (symmetric fp32 column) and headline bench.
# Usage (GPU box): LIBS="ludvm_amd/csrc/libludvm_hip.so _ab/occ4/libludvm_hip.so" bash tools/ab_libs.sh
SIZES="${SIZES:-16384 24576 32768 40960 49152 65536 98304 131072 262144}"
for rep in 1 2; do
  for lib in $LIBS; do
    echo "== $lib sweep $rep"
    LUDVM_HIP_LIB=$lib python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], d['sym_f32_us'], d['sym_f32x2_us'], d.get('direct_f32_us'))"
  done
done
for rep in 1 2; do
  for lib in $LIBS; do
    echo "== $lib bench $rep"
    LUDVM_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --cpu-rows 0 --repeats 1 --cfg4-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
  done
done
