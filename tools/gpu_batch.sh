R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r02_gputests_i.log 2>&1; tail -3 $O/r02_gputests_i.log
python tools/host_boundary_rate.py 2>/dev/null | tee $O/r02_host_boundary_rate.txt
python tools/small_call_latency.py 2>/dev/null | tee $O/r02_small_call_latency.txt
