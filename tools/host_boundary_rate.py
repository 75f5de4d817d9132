"""PCIe-inclusive rate of the host-pointer boundary: LUDVM.induced_velocity-style call with NumPy
float64 arrays (N = 1e6 self-interaction) -- copies, conversions, kernel and copy-back included."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine
n = 1_000_000
rng = np.random.default_rng(20260101)
x, z, g = rng.uniform(-10, 0, n), rng.uniform(-2, 2, n), rng.standard_normal(n) / n
eng = Engine(0)
out = {}
for sym in (1, 0):
    eng.set_symmetric(sym)
    eng.induce(g, x, z, x, z, 0.065)
    t0 = time.perf_counter()
    for _ in range(3):
        u, w = eng.induce(g, x, z, x, z, 0.065)
    el = (time.perf_counter() - t0) / 3
    out["symmetric" if sym else "direct"] = {"s_per_call": el, "pairs_per_s": n * n / el}
print(json.dumps({"host_boundary_N1e6_float64_in_out": out}))
