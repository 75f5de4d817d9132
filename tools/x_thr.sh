#!/bin/bash
for r in 1 2; do for t in 11264 8192 6144 4096; do
  python3 tools/run_configs.py cfg2 --no-timing --sym-threshold $t 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('thr $t', round(d['time_loop_s'],3), d['final_wake'], d['Cl_last'])"
done; done
