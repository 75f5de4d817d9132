#!/bin/bash
# Timeline of marched time steps at the wake sizes given (tools/march_timeline.py): rocprofv3 --kernel-trace per size, then the
# step period, the roll-up kernel and the side chain chord sums -> solve.  Run on the GPU box; writes gpurun_out/$1.
#   bash tools/march_timeline.sh r06_march_timeline_before.txt 8000 16000 24000         [LUDVM_HIP_LIB=... picks the library]
set -o pipefail
OUT=gpurun_out/$1; shift
export TMPDIR=/tmp
W=/tmp/ludvm_mtl_$$; mkdir -p $W gpurun_out
echo "library: ${LUDVM_HIP_LIB:-ludvm_amd/csrc/libludvm_hip.so}" > $OUT
for n in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $W/$n -o t -- python3 tools/march_timeline.py run $n > $W/$n.log 2>&1 || { tail $W/$n.log; exit 1; }
  grep '^{' $W/$n.log >> $OUT
  f=$(find $W/$n -name "*kernel_trace.csv" | head -1)
  python3 tools/march_timeline.py read "$f" $n >> $OUT
done
rm -rf $W
cat $OUT
