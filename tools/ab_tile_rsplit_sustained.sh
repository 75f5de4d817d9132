# Tile size and waves per item of the symmetric kernel on the resident-wake path under SUSTAINED load (SWEEP_SECONDS per
# block, median of three), same box: us per roll-up step.  SYM_T / SYM_R = 0 is the size rule.
SIZES="${SIZES:-16384 20480 24576 32768 40960 49152 65536 81920 98304 131072 196608}"
for v in "0 0" "4 1" "4 2" "4 4" "8 1" "8 2" "8 4"; do set -- $v
  echo "== SYM_T=$1 SYM_R=$2"
  SYM_T=$1 SYM_R=$2 SWEEP_SECONDS=${SWEEP_SECONDS:-0.2} SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 python tools/sweep_rollup.py $SIZES 2>/dev/null | python -c "
import sys, json
print(' '.join('%d:%.1f' % (json.loads(l)['n'], json.loads(l)['sym_f32_us']) for l in sys.stdin if l.startswith('{')))"
done
