# From how many tiles does the quad variant of the symmetric kernel pay?  Resident-wake roll-up under sustained load, same box.
SIZES="${SIZES:-65536 98304 131072 196608 262144 393216 524288}"
for rep in 1 2; do for m in 1024 512 256 128 64; do
  echo "== LUDVM_SYM_QUAD_MIN_TILES=$m pass $rep"
  LUDVM_SYM_QUAD_MIN_TILES=$m SWEEP_SECONDS=${SWEEP_SECONDS:-0.2} SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"
done; done
