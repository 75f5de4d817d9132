#!/bin/bash
# Round 6, side chain of the march: the variants built by tools/build_variant.sh (_ab/libludvm_hip_<name>.so) against round 5's
# library on ONE box -- whole-run fingerprints (same bits?), marched-step timelines, config 2 in alternation.  GPU box.
#   bash tools/r06_chain_ab.sh <names...>         (r05 is always the first)
set -o pipefail
mkdir -p gpurun_out
LIBS="r05 $@"
for L in $LIBS; do
  python3 tools/result_hash.py --timeloop --lib _ab/libludvm_hip_$L.so > gpurun_out/r06_chain_hash_$L.json 2> gpurun_out/r06_chain_hash_$L.err || { tail -5 gpurun_out/r06_chain_hash_$L.err; exit 1; }
  echo "hash $L: $(md5sum < gpurun_out/r06_chain_hash_$L.json | cut -c1-12)"
done
for L in $LIBS; do
  LUDVM_HIP_LIB=$PWD/_ab/libludvm_hip_$L.so bash tools/march_timeline.sh r06_chain_timeline_$L.txt 8000 12000 16000 24000 > /dev/null || exit 1
  grep "^n=" gpurun_out/r06_chain_timeline_$L.txt | sed "s/^/$L /"
done
: > gpurun_out/r06_chain_cfg2_ab.txt
for r in 1 2 3; do
  for L in $LIBS; do
    LUDVM_HIP_LIB=$PWD/_ab/libludvm_hip_$L.so python3 tools/run_configs.py cfg2 --no-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['time_loop_s'],3), d['final_wake'], d['Cl_last'])" | tee -a gpurun_out/r06_chain_cfg2_ab.txt
  done
done
