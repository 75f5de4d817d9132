#!/bin/bash
# Timeline of the kernels of one roll-up step of the resident wake (tools/sweep_rollup.py, symmetric kernel, fp32) at the
# sizes given: rocprofv3 --kernel-trace, then per size the median duration of every kernel of a step and the median gap
# before it (run on the GPU box; writes gpurun_out/$1).
#   bash tools/step_timeline.sh r03_step_timeline.txt 65536 131072 262144
set -o pipefail
OUT=gpurun_out/$1; shift
export TMPDIR=/tmp
W=/tmp/ludvm_tl_$$; mkdir -p $W
: > $OUT
for n in "$@"; do
  SWEEP_SYM_ONLY=1 SWEEP_F32_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $W/$n -o t -- python tools/sweep_rollup.py $n > $W/$n.log 2>&1 || { tail $W/$n.log; exit 1; }
  grep '^{' $W/$n.log >> $OUT
  f=$(find $W/$n -name "*kernel_trace.csv" | head -1)
  python - "$f" $n >> $OUT <<'PY'
import csv, sys, statistics as st
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:60] for r in rows]
# a step = the kernels between two launches of the symmetric pair kernel
idx = [i for i, k in enumerate(names) if 'pair_sym' in k and 'quad' not in k or 'pair_sym_quad' in k]
steps = [(idx[j], idx[j + 1]) for j in range(len(idx) // 2, len(idx) - 1)]
per = {}
period = []
for a, b in steps:
    period.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
    for i in range(a, b):
        gap = (int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
        dur = (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
        per.setdefault((i - a, names[i]), []).append((gap, dur))
print(f"n={sys.argv[2]}: step period median {st.median(period):.1f} us over {len(steps)} steps (under the profiler)")
for (k, name), v in sorted(per.items()):
    if len(v) < len(steps) // 2: continue
    print(f"   [{k}] {name:60s} gap before {st.median(g for g, _ in v):7.1f} us   duration {st.median(d for _, d in v):9.1f} us")
PY
done
rm -rf $W
cat $OUT
