# waves per item (rotation split) of the 512-vortex-tile symmetric kernel on the resident-wake path, mid sizes, same box
for rep in 1 2; do for r in 0 1 2 4; do echo "== SYM_T=8 SYM_R=$r pass $rep"; SYM_T=8 SYM_R=$r SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py ${SIZES:-65536 81920 98304 114688 131072 163840 196608 262144 393216} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['sym_f32_us'], '%.3e' % (d['n']**2 / d['sym_f32_us'] * 1e6))
"; done; done
