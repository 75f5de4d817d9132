# Mixed granularity (LUDVM_SYM_MIXED=1: bulk by single waves, the last chip-load of items by four waves each) against the
# size rule, resident-wake roll-up, sustained blocks, same box: n:us per step
SIZES="${SIZES:-20480 24576 28672 32768 36864 40960 45056 49152 53248 57344 61440 65536 73728 81920 98304 114688 131072}"
for rep in 1 2; do for m in 0 1; do echo "== LUDVM_SYM_MIXED=$m pass $rep"; LUDVM_SYM_MIXED=$m SWEEP_SECONDS=0.2 SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"; done; done
