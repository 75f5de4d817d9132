#!/bin/bash
# Kernel statistics + the three PMC passes of ONE python command, from one box (run on the GPU box):
#   bash tools/profile_cmd.sh <name> <script.py> [args...]       ->  gpurun_out/<name>_{kernel_stats,pmc_sq,pmc_fetch,pmc_write}.csv
# rocprofv3 --kernel-trace --stats of `python3 <script.py> args`, then separate --pmc passes of the same command (SQ
# counters; FETCH_SIZE; WRITE_SIZE + atomics: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") with
# LUDVM_PROFILE_PMC=1 exported (scripts shorten their repetitions under it).  The interpreter itself follows `--`: no
# env / shell / launcher hop.  Counter files are reduced to per-kernel means (kernel,counter,dispatches,mean_per_dispatch).
set -o pipefail
NAME=$1; shift
P=gpurun_out/$NAME
export TMPDIR=/tmp
W=/tmp/ludvm_prof_$$; mkdir -p $W gpurun_out
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $W/ks -o ks -- python3 "$@" > ${P}_under_rocprof.out 2>$W/ks.err || { tail $W/ks.err; exit 1; }
find $W/ks -name "*kernel_stats.csv" -exec cp {} ${P}_kernel_stats.csv \;
rm -rf $W/ks
export LUDVM_PROFILE_PMC=1
pmc() {  # name counters...
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $W/$name -o p -- python3 "$@" > $W/$name.out 2>$W/$name.err || { tail $W/$name.err; return 1; }
  find $W/$name -name "*counter_collection.csv" -exec cp {} $W/${name}_raw.csv \;
  python3 - "$W/${name}_raw.csv" "${P}_pmc_${name}.csv" <<'PY'
import csv, collections, sys
acc = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        acc.setdefault((r['Kernel_Name'].split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), v in acc.items():
        f.write(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.6g}\n")
print(sys.argv[2], [(k[:40], c, round(sum(v) / len(v))) for (k, c), v in acc.items() if 'pair_' in k][:12])
PY
  rm -rf $W/$name $W/${name}_raw.csv
}
pmc sq SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- "$@" || exit 1
pmc fetch FETCH_SIZE -- "$@" || exit 1
pmc write WRITE_SIZE TCC_EA0_ATOMIC_sum -- "$@" || exit 1
rm -rf $W
head -6 ${P}_kernel_stats.csv | cut -c1-220
