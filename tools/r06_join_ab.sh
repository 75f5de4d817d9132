#!/bin/bash
# Round 6: the flag join of an overlapped march step (MarchState::solve_step) against the event join, ONE box: whole-run
# fingerprints (same bits?), marched-step timelines, config 2 in alternation.   bash tools/r06_join_ab.sh <libA> <libB>   (names in _ab/)
set -o pipefail
mkdir -p gpurun_out
for L in "$@"; do
  python3 tools/result_hash.py --timeloop --lib _ab/libludvm_hip_$L.so > gpurun_out/r06_join_hash_$L.json 2> gpurun_out/r06_join_hash_$L.err || { tail -5 gpurun_out/r06_join_hash_$L.err; exit 1; }
  echo "hash $L: $(md5sum < gpurun_out/r06_join_hash_$L.json | cut -c1-12)"
done
for L in "$@"; do
  LUDVM_HIP_LIB=$PWD/_ab/libludvm_hip_$L.so bash tools/march_timeline.sh r06_join_timeline_$L.txt 12000 16000 24000 48000 > /dev/null || exit 1
  grep -v "^{" gpurun_out/r06_join_timeline_$L.txt | grep "^n=\|march_finish_sym\|pair_sym" | sed "s/^/$L /"
done
: > gpurun_out/r06_join_cfg2_ab.txt
for r in 1 2 3; do
  for L in "$@"; do
    LUDVM_HIP_LIB=$PWD/_ab/libludvm_hip_$L.so python3 tools/run_configs.py cfg2 --no-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['time_loop_s'],3), d['final_wake'], d['Cl_last'])" | tee -a gpurun_out/r06_join_cfg2_ab.txt
  done
done
