"""Symmetric kernel: sensitivity to the number of d-chunks at mid-size N, interleaved rounds (run on the GPU box)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_symmetric(2)
rng = np.random.default_rng(1)
for n in (24576, 32768, 49152, 65536, 90000):
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    xo, zo = torch.empty_like(x), torch.empty_like(z)
    yss = (16, 32, 64, 128, 256)
    best = {ys: 1e9 for ys in yss}
    def run(ys, reps):
        eng.set_tuning(0, ys)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            eng.advect_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, n, 0.065, 1e-3, xo.data_ptr(), zo.data_ptr())
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    for ys in yss: run(ys, 20)
    for rnd in range(6):
        for ys in yss:
            best[ys] = min(best[ys], run(ys, 30))
    print(json.dumps({"n": n, **{"ys%d" % ys: float("%.3e" % (n * n / best[ys])) for ys in yss}}), flush=True)
