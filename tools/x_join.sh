#!/bin/bash
set -o pipefail
for L in r06now x_nojoin; do
  LUDVM_HIP_LIB=$PWD/_ab/libludvm_hip_$L.so bash tools/march_timeline.sh x_join_timeline_$L.txt 24000 32000 48000 > /dev/null || exit 1
  grep -v "^{" gpurun_out/x_join_timeline_$L.txt | sed "s/^/$L /"
done
for r in 1 2; do for L in r06now x_nojoin; do
  LUDVM_HIP_LIB=$PWD/_ab/libludvm_hip_$L.so python3 tools/run_configs.py cfg2 --no-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['time_loop_s'],3), d['final_wake'], d['Cl_last'])"
done; done
