"""fp64 roll-up step (resident wake, ludvm_wake_advect) at small wake sizes (run on the GPU box)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LUDVM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ludvm_amd", "csrc", "libludvm_hip_exp.so"))  # measurement build: forced variants / A-B switches
from ludvm_amd import Engine
eng = Engine(0)
rng = np.random.default_rng(2)
out = {"LUDVM_SMALL_TILE_MAX_F64": os.environ.get("LUDVM_SMALL_TILE_MAX_F64", "default")}
fx, fz, fg = rng.uniform(-1, 0, 80), rng.uniform(-0.1, 0.1, 80), rng.standard_normal(80) * 1e-2
for n in (300, 600, 1200, 2400, 4096, 8192, 16384):
    eng.wake_clear()
    eng.wake_append(rng.uniform(-10, 0, n), rng.uniform(-2, 2, n), rng.standard_normal(n) / n)
    for _ in range(5):
        eng.wake_advect(1e-6, fx, fz, fg, 0.065, precision="f64")
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.wake_advect(1e-6, fx, fz, fg, 0.065, precision="f64")
    eng.synchronize()
    out[str(n)] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
print(json.dumps(out))
