#!/bin/bash
# Run length (d-chunks) of the XCD rotation of the symmetric kernels (LUDVM_XCD_RUN; 0 = a launch's chunks / 8), same box, alternating:
# headline call, config 2, and the FETCH_SIZE of the headline launch (rocprofv3 --pmc, separate pass).
RUNS="${RUNS:-0 1 2 4 1000}"
for rep in 1 2 3; do for k in $RUNS; do echo "== LUDVM_XCD_RUN=$k bench $rep"
  LUDVM_XCD_RUN=$k python bench.py --steps 10 --warmup 3 --cpu-rows 0 --repeats 1 --cfg4-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
for rep in 1 2; do for k in $RUNS; do echo "== LUDVM_XCD_RUN=$k config 2, $rep"
  LUDVM_XCD_RUN=$k python tools/run_configs.py cfg2 --no-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],3), d['final_wake'], d['Cl_last'])"
done; done
export TMPDIR=/tmp
for k in $RUNS; do W=/tmp/xr_$k; rm -rf $W
  LUDVM_XCD_RUN=$k timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W -o p -- python bench.py --steps 2 --warmup 1 --repeats 0 --cpu-rows 0 --cfg4-steps 0 > /dev/null 2>$W.err || tail -3 $W.err
  f=$(find $W -name "*counter_collection.csv" | head -1)
  python - "$f" $k <<'PY'
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'pair_sym_quad' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print(f"== LUDVM_XCD_RUN={sys.argv[2]} FETCH_SIZE of the quad launch, KB (mean of {len(v)}): {sum(v) / len(v):.0f}")
PY
done
