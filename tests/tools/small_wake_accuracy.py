#!/usr/bin/env python3
"""fp32 roll-up accuracy of SMALL wakes (tens to hundreds of vortices, alternating TEV / LEV order near the airfoil) against
the float64 C oracle, symmetric kernel forced from 8 vortices and direct kernel (GPU box; test infrastructure).
Usage: python tests/tools/small_wake_accuracy.py   (works in older trees too: only Engine calls)"""
import os
import sys

import numpy as np

ROOT = os.environ.get("LUDVM_TREE") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ludvm_amd import Engine  # noqa: E402
from oracle import c_oracle  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(4)
for n in (9, 33, 60, 137, 275, 513, 700, 1500):
    m = n // 2
    xt = -0.05 - np.arange(n - m) * 5e-3
    xl = -1.0 - np.arange(m) * 5e-3
    x, z = np.empty(n), np.empty(n)
    x[0::2], x[1::2] = xt, xl
    z[0::2] = 0.05 * np.sin(7 * xt) + 1e-3 * rng.standard_normal(n - m)
    z[1::2] = 0.1 + 0.05 * np.cos(5 * xl) + 1e-3 * rng.standard_normal(m)
    g = rng.standard_normal(n) * 1e-2
    vc = 1.3 * 5e-3
    ur, wr = c_oracle.induced_velocity(g, x, z, x, z, vc)
    scale = max(np.abs(ur).max(), np.abs(wr).max())
    row = []
    for mode in (8, 0):
        eng.set_symmetric(mode)
        for prec in ("f32", "f32x2"):
            eng.wake_clear()
            eng.wake_append(x, z, g)
            u, w = eng.wake_advect(1e-3, [], [], [], vc, precision=prec, return_velocity=True)
            row.append(max(np.abs(u - ur).max(), np.abs(w - wr).max()) / scale)
    print(f"n = {n:5d}: symmetric f32 {row[0]:.2e}  f32x2 {row[1]:.2e}   direct f32 {row[2]:.2e}  f32x2 {row[3]:.2e}")
