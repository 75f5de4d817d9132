"""fp64 parity-mode throughput of the roll-up (direct fp64 kernel), run on the GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine
eng = Engine(0)
rng = np.random.default_rng(3)
for n in (50000, 200000):
    x, z, g = rng.uniform(-10, 0, n), rng.uniform(-2, 2, n), rng.standard_normal(n) / n
    eng.wake_clear(); eng.wake_append(x, z, g)
    eng.wake_advect(1e-3, [], [], [], 0.065, precision="f64"); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.wake_advect(1e-3, [], [], [], 0.065, precision="f64")
    eng.synchronize()
    el = (time.perf_counter() - t0) / 3
    print(json.dumps({"n": n, "fp64_rollup_pairs_per_s": float("%.3e" % (n * n / el)), "ms": round(el * 1e3, 2)}))
