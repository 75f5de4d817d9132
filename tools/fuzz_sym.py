"""One-off fuzz: symmetric vs direct self-interaction over random sizes (run on the GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(12345)
worst = 0.0
sizes = ([16384, 16385, 16639, 16640, 40959, 40960, 40961, 41471, 41472, 98303, 98304, 98305, 98815, 98816, 99328]
         + [int(v) for v in rng.integers(16384, 100000, 40)] + [int(v) for v in rng.integers(100000, 400000, 20)])
for n in sizes:
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    out = {}
    for mode in (0, 1):
        eng.set_symmetric(mode)
        u = torch.full((n,), float("nan"), device=dev); w = torch.full((n,), float("nan"), device=dev)
        eng.induce_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, x.data_ptr(), z.data_ptr(), n, 0.065, u.data_ptr(), w.data_ptr())
        torch.cuda.synchronize()
        out[mode] = (u, w)
    scale = max(float(out[0][0].abs().max()), float(out[0][1].abs().max()))
    err = max(float((out[0][0] - out[1][0]).abs().max()), float((out[0][1] - out[1][1]).abs().max())) / scale
    assert torch.isfinite(out[1][0]).all() and torch.isfinite(out[1][1]).all(), n
    worst = max(worst, err)
    if err > 1e-5:
        print("MISMATCH", n, err)
print("sizes", len(sizes), "worst rel diff", worst)
