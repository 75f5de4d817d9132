#!/bin/bash
# Round 4, VERDICT item 2: kernel statistics + counter passes of every dominant kernel on the current code, one box.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== cfg3 symmetric"; bash tools/profile_bench.sh r04_bench_cfg3_sym || exit 1
echo "== cfg3 direct";    bash tools/profile_bench.sh r04_bench_cfg3_direct --symmetric 0 --cfg4-steps 0 || exit 1
echo "== cfg5";           bash tools/profile_cmd.sh r04_config5 tools/run_configs.py cfg5 || exit 1
echo "== cfg2 sizes";     STEPS=150 bash tools/profile_cmd.sh r04_config2_sizes tools/march_step_overhead.py 36000 60000 || exit 1
echo "== cfg2 full";      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2ks -o ks -- python3 tools/run_configs.py cfg2 > gpurun_out/r04_config2_under_rocprof.json 2>/tmp/c2.err || { tail /tmp/c2.err; exit 1; }
find /tmp/c2ks -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_config2_kernel_stats.csv \; ; rm -rf /tmp/c2ks
python3 tools/run_configs.py cfg2 > gpurun_out/r04_config2.json 2>/dev/null || exit 1
echo "== cfg4 one rank, full size, library collectives"
timeout -k 10 900 python3 bench.py --workload cfg4 --steps 3 --warmup 1 --collectives library --budget-s 500 --cpu-rows 0 > gpurun_out/r04_cfg4_one_rank_library.json 2>gpurun_out/r04_cfg4_one_rank_library.err || { tail gpurun_out/r04_cfg4_one_rank_library.err; exit 1; }
cat gpurun_out/r04_config2.json; cut -c1-1500 gpurun_out/r04_cfg4_one_rank_library.json
