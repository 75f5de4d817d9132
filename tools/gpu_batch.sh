R=$GRAFT_REPO_ROOT
LUDVM_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29571 $R/tools/dist_class_check.py > $R/gpurun_out/r02_dist_check.log 2>&1
tail -40 $R/gpurun_out/r02_dist_check.log
python -m pytest $R/tests/test_gpu_kernel.py -m gpu -q -k "far_wake or patch" 2>&1 | tail -5
