# d-chunks per tile (ludvm_set_tuning's second argument; 0 = the rule) of the plain symmetric kernel on the resident-wake
# path, sustained blocks, same box: n:us per roll-up step
SIZES="${SIZES:-131072 196608 262144 393216 500000}"
for rep in 1 2; do for ys in ${YSS:-0 32 128 256}; do echo "== SYM_YS=$ys pass $rep"; SYM_YS=$ys SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"; done; done
