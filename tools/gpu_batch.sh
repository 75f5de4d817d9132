R=$GRAFT_REPO_ROOT
for cfg in "0 0" "8 2" "8 4" "4 0"; do set -- $cfg; echo "T=$1 R=$2"; SYM_T=$1 SYM_R=$2 python $R/tools/sweep_rollup.py 16384 32768 40960 49152 65536 98304 2>/dev/null | cut -c1-60; done
python $R/tools/run_configs.py cfg2 --precision f32 --no-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW tf50 wall', d['wall_s'], d['final_wake'])"
python $R/_old_r1/tools/run_configs.py cfg2 --precision f32 --no-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OLD tf50 wall', d['wall_s'], d['final_wake'])"
