for rep in 1 2; do for ys in 0 8 16 32 128; do echo "== SYM_YS=$ys pass $rep"; SYM_YS=$ys SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py 49152 65536 98304 131072 196608 262144 393216 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['sym_f32_us'], '%.3e' % (d['n']**2 / d['sym_f32_us'] * 1e6))
"; done; done
