#!/bin/bash
# SQ counters of the headline kernel for this tree and for the tree under _ab/r2, same box (GPU box).
export TMPDIR=/tmp
CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS"
run() {  # dir label
  (cd $1 && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_$2 -o p -- python bench.py --steps 3 --warmup 1 --repeats 0 --cpu-rows 0 $3 > /dev/null 2>/tmp/pmc_$2.err) || tail /tmp/pmc_$2.err
  python - /tmp/pmc_$2 $2 <<'PY'
import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'pair_sym' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print(sys.argv[2], {k: f"{sum(v)/len(v):.5g}" for k, v in acc.items()})
PY
}
run _ab/r2 r2 ""
run . new "--cfg4-steps 0"
run _ab/r2 r2 ""
run . new "--cfg4-steps 0"
