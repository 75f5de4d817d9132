"""Symmetric kernel: sensitivity to the number of d-chunks per tile (ludvm_set_tuning's second argument; 0 = the size rule)
at mid-size N, interleaved rounds, best of six (run on the GPU box).
    python tools/sweep_ysplit.py [sizes...]      YSS=0,64,128,256 chooses the list"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LUDVM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ludvm_amd", "csrc", "libludvm_hip_exp.so"))  # measurement build: forced variants / A-B switches
from ludvm_amd import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_symmetric(2)
rng = np.random.default_rng(1)
sizes = [int(a) for a in sys.argv[1:]] or [65536, 98304, 131072, 196608, 262144, 393216]
for n in sizes:
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    xo, zo = torch.empty_like(x), torch.empty_like(z)
    yss = tuple(int(v) for v in os.environ.get('YSS', '0,32,64,96,128,192,256,512').split(','))
    best = {ys: 1e9 for ys in yss}
    def run(ys, reps):
        eng.set_tuning(0, ys)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            eng.advect_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, n, 0.065, 1e-3, xo.data_ptr(), zo.data_ptr())
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    reps = max(3, min(30, int(2e10 / (n * n))))
    for ys in yss: run(ys, reps)
    for rnd in range(6):
        for ys in yss:
            best[ys] = min(best[ys], run(ys, reps))
    print(json.dumps({"n": n, **{"ys%d" % ys: float("%.3e" % (n * n / best[ys])) for ys in yss}}), flush=True)
