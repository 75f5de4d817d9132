#!/bin/bash
# The headline bench line and its profiles from ONE box (run on the GPU box; writes gpurun_out/$1_*):
#   bench line; rocprofv3 --kernel-trace --stats of the same command (without the legs of the other configs: config 4, 5, 2, 1
#   have profiles of their own); three separate --pmc passes (SQ, FETCH_SIZE,
#   WRITE_SIZE + atomics: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter files are reduced
#   to per-kernel means (kernel,counter,dispatches,mean_per_dispatch).
# Usage: bash tools/profile_bench.sh r03_final [extra bench.py arguments]
set -o pipefail
P=gpurun_out/$1; shift
EXTRA="$@"
export TMPDIR=/tmp
W=/tmp/ludvm_prof_$$; mkdir -p $W
timeout -k 10 600 python bench.py --steps 20 --warmup 5 $EXTRA > ${P}_bench.json 2>${P}_bench.err || { tail ${P}_bench.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/ks -o ks -- python bench.py --steps 20 --warmup 5 --cfg4-steps 0 --cfg5 0 --cfg2 0 --cfg1 0 $EXTRA > ${P}_bench_under_rocprof.json 2>$W/ks.err || { tail $W/ks.err; exit 1; }
find $W/ks -name "*kernel_stats.csv" -exec cp {} ${P}_kernel_stats.csv \;
pmc() {  # name counters...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $W/$name -o p -- python bench.py --steps 3 --warmup 1 --repeats 0 --cpu-rows 0 --cfg4-steps 0 --cfg5 0 --cfg2 0 --cfg1 0 $EXTRA > $W/$name.json 2>$W/$name.err || { tail $W/$name.err; return 1; }
  find $W/$name -name "*counter_collection.csv" -exec cp {} $W/${name}_raw.csv \;
  python - "$W/${name}_raw.csv" "${P}_pmc_${name}.csv" <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    acc.setdefault((r['Kernel_Name'].split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), v in acc.items():
        f.write(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.6g}\n")
print(sys.argv[2], [(c, round(sum(v) / len(v))) for (k, c), v in acc.items() if 'pair_' in k])
PY
}
pmc sq SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE || exit 1
pmc fetch FETCH_SIZE || exit 1
pmc write WRITE_SIZE TCC_EA0_ATOMIC_sum || exit 1
rm -rf $W
python - <<PY
import json
for f in ("${P}_bench.json", "${P}_bench_under_rocprof.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d.get('config4_one_gpu', {}).get('value'))
print(open("${P}_kernel_stats.csv").read().splitlines()[1][:200])
PY
