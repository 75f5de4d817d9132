#!/bin/bash
# Kernel statistics + counter passes of every dominant kernel on the current code, one box -- config 2's rows from config 2's
# OWN full run (no proxy; round 5).  Run on the GPU box:  bash tools/profile_batch.sh <round, e.g. r06> [stage ...]
# Stages: cfg2, cfg3sym, cfg3direct, cfg5.  Writes gpurun_out/<round>_*; copy to profiles/ (tools/roofline_table.py <round> reads them).
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$1; shift
STAGES=${@:-cfg2 cfg3sym cfg3direct cfg5}
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2 > gpurun_out/${R}_box_clocks.txt
for st in $STAGES; do
  case $st in
  cfg2)
    echo "== cfg2 pilot: one counter pass over the first 5000 steps (25 000 dispatches), timed"
    t0=$(date +%s)
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/c2pilot -o p -- python3 tools/run_configs.py cfg2 --tf 5 > gpurun_out/${R}_config2_pilot.json 2>/tmp/c2pilot.err || { tail /tmp/c2pilot.err; exit 1; }
    t1=$(date +%s); rm -rf /tmp/c2pilot
    echo "pilot: $((t1 - t0)) s for 5000 steps under --pmc" | tee gpurun_out/${R}_config2_pilot_time.txt
    if [ $((t1 - t0)) -gt 75 ]; then echo "full-run counter passes would not fit; stopping here"; exit 3; fi
    echo "== cfg2 full run: kernel statistics + three counter passes"
    bash tools/profile_cmd.sh ${R}_config2 tools/run_configs.py cfg2 || exit 1
    python3 tools/run_configs.py cfg2 > gpurun_out/${R}_config2.json 2>/dev/null || exit 1
    cut -c1-600 gpurun_out/${R}_config2.json ;;
  cfg3sym)    echo "== cfg3 symmetric"; bash tools/profile_bench.sh ${R}_bench_cfg3_sym || exit 1 ;;
  cfg3direct) echo "== cfg3 direct";    bash tools/profile_bench.sh ${R}_bench_cfg3_direct --symmetric 0 --cfg4-steps 0 --cfg5 0 --cfg2 0 --cfg1 0 || exit 1 ;;
  cfg5)       echo "== cfg5";           bash tools/profile_cmd.sh ${R}_config5 tools/run_configs.py cfg5 || exit 1 ;;
  esac
done
