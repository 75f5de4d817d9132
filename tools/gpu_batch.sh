R=$GRAFT_REPO_ROOT
cd $R
python _old_r1/bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-rows 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OLD cfg4', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-rows 0 --repeats 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW cfg4', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-rows 0 --repeats 0 --splits 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW cfg4 splits 8', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-rows 0 --repeats 0 --splits 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW cfg4 splits 64', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
python bench.py --steps 5 --warmup 2 --cpu-rows 0 --repeats 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW cfg3', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
