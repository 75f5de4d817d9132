R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
: > $O/r02_fuzz_march_seeds_1_to_7.txt
for seed in 1 2 3 4 5 6 7; do python $R/tools/fuzz_march.py --cases 24 --seed $seed >> $O/r02_fuzz_march_seeds_1_to_7.txt 2>/dev/null; echo "seed $seed rc $?" >> $O/r02_fuzz_march_seeds_1_to_7.txt; done
grep -E "failures|rc" $O/r02_fuzz_march_seeds_1_to_7.txt
echo SWEEP; python $R/tools/sweep_rollup.py 11264 16384 24576 32768 40960 49152 65536 98304 131072 262144 > $O/r02_sweep_rollup.txt 2>/dev/null; cat $O/r02_sweep_rollup.txt
python $R/tools/sweep_n.py 16384 65536 262144 1048576 > $O/r02_sweep_n.txt 2>/dev/null; cat $O/r02_sweep_n.txt
