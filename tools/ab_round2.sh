#!/bin/bash
# Same-box A/B of this tree against the round-2 tree unpacked (and built) under _ab/r2: roll-up sweep and the headline bench,
# alternating.  Usage (GPU box): bash tools/ab_round2.sh > gpurun_out/ab.txt
SIZES="${SIZES:-16384 24576 32768 40960 49152 65536 98304 131072 262144}"
for rep in 1 2; do
  echo "== round-2 tree, sweep $rep"; (cd _ab/r2 && python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], d['sym_f32_us'], d['sym_f32x2_us'], d.get('direct_f32_us'))")
  echo "== this tree, sweep $rep"; python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], d['sym_f32_us'], d['sym_f32x2_us'], d.get('direct_f32_us'))"
done
for rep in 1 2; do
  echo "== round-2 tree, bench $rep"; (cd _ab/r2 && python bench.py --steps 10 --warmup 2 --cpu-rows 0 --repeats 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])")
  echo "== this tree, bench $rep"; python bench.py --steps 10 --warmup 2 --cpu-rows 0 --repeats 1 --cfg4-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
done
