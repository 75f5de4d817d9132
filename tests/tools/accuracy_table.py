#!/usr/bin/env python3
"""Measured fp32 accuracy of the pair sums against the float64 C oracle (run on the GPU box; test infrastructure: it
uses the oracle).  Prints the table DESIGN.md section 3.1c / 2 quotes:
  1. one roll-up of a late-time wake (|x| ~ 50, spacing 1e-3, v_core = 1.3e-3): single sheet, alternating TEV / LEV order,
     and the wake a real config-2 run leaves -- per kernel (symmetric 512 / 256 tile, direct) and precision;
  2. the fp32 flow field at config 5's dr on a 64 x 64 patch: velocity and vorticity against the oracle's float64 fields."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CONFIG1  # noqa: E402
from ludvm_amd import LUDVM, Engine  # noqa: E402
from oracle import c_oracle, ludvm_oracle as O  # noqa: E402
from test_gpu_wake import _late_time_wake  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(31)
vc = 1.3e-3


def rollup_errors(x, z, g, vcore, label, kernels):
    ur, wr = c_oracle.induced_velocity(g, x, z, x, z, vcore)
    scale = max(np.abs(ur).max(), np.abs(wr).max())
    for name, mode, tile in kernels:
        eng.set_symmetric(mode)
        eng.set_sym_tuning(tile, 0)
        row = []
        for prec in ("f32", "f32x2"):
            eng.wake_clear()
            eng.wake_append(x, z, g)
            u, w = eng.wake_advect(1e-3, [], [], [], vcore, precision=prec, return_velocity=True)
            row.append(max(np.abs(u - ur).max(), np.abs(w - wr).max()) / scale)
        print(f"  {label:34s} {name:22s} f32 (local origins) {row[0]:.2e}   f32x2 (hi+lo) {row[1]:.2e}")
    eng.set_symmetric(1)
    eng.set_sym_tuning(0, 0)


print("1. one roll-up, max error / max|u| against the float64 C oracle")
K = (("symmetric, 512 tile", 1, 8), ("symmetric, 256 tile", 1, 4), ("direct", 0, 0))
for layout in ("sheet", "interleaved"):
    n = 40000
    x, z = _late_time_wake(layout, n, rng)
    g = rng.standard_normal(n) * 1e-3
    rollup_errors(x, z, g, vc, f"synthetic {layout}, n = {n}", K)
sim = LUDVM(**dict(CONFIG1, dt=1e-3, tf=2.6), verbose=False, engine=eng, precision="f64", history="sparse")
n = eng.wake_size()
x, z, g = eng.wake_read(0, n, gamma=True)
rollup_errors(x - 47.0, z, g, sim.v_core, f"config-2 run at step 2600, n = {n}", (("symmetric, 256 tile", 1024, 4), ("direct", 0, 0)))

print("2. fp32 flow field, 64 x 64 patch of config 5's grid (dr = 8/4096) in a 1e5-vortex synthetic wake")
rng = np.random.default_rng(20260101)
n = 100_000
xs, zs, g = rng.uniform(-10, 0, n), rng.uniform(-2, 2, n), rng.standard_normal(n) / n
dr, nx, nz = 8.0 / 4096, 64, 64
xmin, zmin = -8.0 + 1800 * dr, -4.0 + 2100 * dr
X, Z = np.meshgrid(xmin + np.arange(nx) * dr, zmin + np.arange(nz) * dr, indexing="ij")
ur, wr = c_oracle.induced_velocity(g, xs, zs, X.ravel(), Z.ravel(), 0.065)
ur, wr = ur.reshape(nx, nz), wr.reshape(nx, nz)
orr = O.vorticity(ur[None], wr[None], X, Z)[0]
for prec in ("f32", "f64"):
    u, w, ome = eng.flowfield_vorticity(xmin, zmin, dr, nx, nz, g, xs, zs, 0.065, precision=prec)
    ev = max(np.abs(u - ur).max(), np.abs(w - wr).max()) / max(np.abs(ur).max(), np.abs(wr).max())
    eo = np.abs(ome - orr).max() / np.abs(orr).max()
    print(f"  precision {prec}: velocity {ev:.2e} of max|u|, vorticity {eo:.2e} of max|omega|")
