SIZES="49152 57344 65536 73728 81920 98304 114688 131072 163840 196608"
for t in 768 1536 3072 6144; do echo "== LUDVM_SYM_MIXED=1 LUDVM_SYM_TAIL_ITEMS=$t"; LUDVM_SYM_TAIL_ITEMS=$t LUDVM_SYM_MIXED=1 SWEEP_SECONDS=0.2 SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"; done
echo "== rule"; SWEEP_SECONDS=0.2 SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"
