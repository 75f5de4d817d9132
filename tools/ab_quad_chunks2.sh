# Tapered d-chunks of the quad variant (the rule) against uniform ones (--splits k), headline call, same box, alternating
b() { python bench.py --steps 10 --warmup 3 --cpu-rows 0 --repeats 1 --cfg4-steps 0 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"; }
for rep in 1 2 3; do for s in 64 128 0; do echo "== --splits $s pass $rep"; b --splits $s; done; done
