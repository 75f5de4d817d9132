set -x
python -m pytest tests -m gpu -q > gpurun_out/r02_gputests_d.log 2>&1; tail -8 gpurun_out/r02_gputests_d.log
python -m pytest tests/test_gpu_kernel.py -m gpu -q -s -k integration_md 2>&1 | grep "binding, precision"
python tools/t2_windows.py > gpurun_out/r02_t2_windows.txt 2>&1
python bench.py --steps 10 --warmup 3 --cpu-rows 0 > gpurun_out/r02_bench_cfg3_sym.json 2> gpurun_out/r02_bench_cfg3_sym.err; cat gpurun_out/r02_bench_cfg3_sym.json
python tools/run_configs.py cfg2 --precision f32 --no-timing > gpurun_out/r02_cfg2_f32.json 2>&1; cat gpurun_out/r02_cfg2_f32.json
python tools/run_configs.py cfg2 --precision f32x2 --no-timing > gpurun_out/r02_cfg2_f32x2.json 2>&1; cat gpurun_out/r02_cfg2_f32x2.json
