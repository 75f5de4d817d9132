"""Direct vs symmetric self-advection step (ludvm_advect_dev_f32) over N: where should the symmetric
kernel take over, and how efficient are mid-size launches?  Run on the GPU box."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
rows = []
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 65536, 131072, 262144]
for n in sizes:
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    xo, zo = torch.empty_like(x), torch.empty_like(z)
    res = {"n": n}
    for name, mode in (("direct", 0), ("symmetric", 2)):
        eng.set_symmetric(mode)
        reps = max(3, min(200, int(2e10 / (n * n))))
        for _ in range(3):
            eng.advect_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, n, 0.065, 1e-3, xo.data_ptr(), zo.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.advect_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, n, 0.065, 1e-3, xo.data_ptr(), zo.data_ptr())
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        res[name + "_us"] = round(el * 1e6, 1)
        res[name + "_pairs_per_s"] = float("%.3e" % (n * n / el))
    rows.append(res)
    print(json.dumps(res), flush=True)
