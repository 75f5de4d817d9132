"""Symmetric kernel: sensitivity to the number of d-chunks (ysplit) at mid-size N.  Run on the GPU box."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_symmetric(2)
rng = np.random.default_rng(1)
for n in (16384, 32768, 49152, 65536, 131072, 1000000):
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    xo, zo = torch.empty_like(x), torch.empty_like(z)
    res = {"n": n}
    for ys in (64, 64, 96, 128, 192, 256, 512):
        eng.set_tuning(0, ys)
        reps = max(3, min(100, int(1e10 / (n * n)) + 3))
        for _ in range(2):
            eng.advect_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, n, 0.065, 1e-3, xo.data_ptr(), zo.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.advect_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, n, 0.065, 1e-3, xo.data_ptr(), zo.data_ptr())
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        res.setdefault("ys%d" % ys, []).append(float("%.3e" % (n * n / el)))
    print(json.dumps(res), flush=True)
