#!/bin/bash
# Same-box A/B of this tree against the round-2 tree (built under _ab/r2): BASELINE config 2 (wall seconds) and the headline call.
for rep in 1 2 3; do
  echo "== round-2 tree, config 2, $rep"; (cd _ab/r2 && python tools/run_configs.py cfg2 --no-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],3), d['final_wake'])")
  echo "== this tree, config 2, $rep"; python tools/run_configs.py cfg2 --no-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],3), d['final_wake'])"
done
for rep in 1 2 3; do
  echo "== round-2 tree, bench $rep"; (cd _ab/r2 && python bench.py --steps 10 --warmup 3 --cpu-rows 0 --repeats 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")
  echo "== this tree, bench $rep"; python bench.py --steps 10 --warmup 3 --cpu-rows 0 --repeats 1 --cfg4-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
