# Same-box A/B/C of the headline call (N = 1e6): round-2 tree (_ab/r2), this tree without and with the quad variant.
b() { python bench.py --steps 10 --warmup 2 --cpu-rows 0 --repeats 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"; }
for rep in 1 2 3; do
  echo "== round-2 tree $rep"; (cd _ab/r2 && b)
  echo "== this tree, LUDVM_SYM_QUAD=0 $rep"; LUDVM_SYM_QUAD=0 b --cfg4-steps 0
  echo "== this tree $rep"; b --cfg4-steps 0
done
