b() { python bench.py --steps 10 --warmup 3 --cpu-rows 0 --repeats 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d.get('config4_one_gpu') or {}; print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], c.get('value'), c.get('ms_per_step'))"; }
for rep in 1 2; do echo "== --splits 64 (old rule) $rep"; b --splits 64 --cfg4-steps 0; echo "== new rule $rep"; b --cfg4-steps 0; done
echo "== new rule with config 4 on one GPU"; b --cfg4-steps 2
