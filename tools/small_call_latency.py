#!/usr/bin/env python3
"""Latency of the stateless host call (ludvm_induce_f64 through Engine.induce) at the shapes of a README-size run."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(3)
out = {"lib": os.environ.get("LUDVM_HIP_LIB", "default")}
for nt, ns in ((80, 1), (80, 600), (400, 680), (2000, 3000)):
    xs, zs, g = rng.uniform(-10, 0, ns), rng.uniform(-2, 2, ns), rng.standard_normal(ns)
    xt, zt = rng.uniform(-10, 0, nt), rng.uniform(-2, 2, nt)
    for prec in ("f32", "f64"):
        for _ in range(50):
            eng.induce(g, xs, zs, xt, zt, 0.065, precision=prec)
        best = 1e9
        for rnd in range(5):
            t0 = time.perf_counter()
            for _ in range(400):
                eng.induce(g, xs, zs, xt, zt, 0.065, precision=prec)
            best = min(best, (time.perf_counter() - t0) / 400)
        out[f"{nt}x{ns}_{prec}_us"] = round(best * 1e6, 1)
print(json.dumps(out))
