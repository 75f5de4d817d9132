#!/usr/bin/env python3
"""Calibration of the sparse rule (ctx.hpp, too_sparse: 150 v_core for a reordered cloud, 300 for a set compact as given) [GPU box; uses the oracle as the checker]: fp32 on
local origins in Morton order -- through the resident wake, which has no hi+lo fallback of its own -- for clouds of several
densities and cores: max error on 512 sampled targets / their max|u| against (mean class extent) / v_core.
    python tests/tools/extent_rule_calibration.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ludvm_amd import Engine  # noqa: E402
from oracle import c_oracle  # noqa: E402

eng = Engine(0)
cases = [(200_000, 1.3e-3), (500_000, 1.3e-3), (1_000_000, 1.3e-3), (2_000_000, 1.3e-3), (1_000_000, 6.5e-4), (1_000_000, 2.6e-3),
         (1_000_000, 1.3e-2), (100_000, 1.3e-2), (100_000, 0.065)]
for n, vc in cases:
    rng = np.random.default_rng(n + int(vc * 1e6))
    x, z, g = rng.uniform(-60.0, -50.0, n), rng.uniform(-2.0, 2.0, n), rng.standard_normal(n) * 1e-2
    sel = rng.choice(n, 512, replace=False)
    ur, wr = c_oracle.induced_velocity(g, x, z, x[sel], z[sel], vc)
    scale = max(np.abs(ur).max(), np.abs(wr).max())
    order, reordered, extent = eng.spatial_order(x, z, with_extent=True)
    slot = np.empty(n, np.int64)
    slot[order] = np.arange(n)
    rec = {"n": n, "v_core": vc, "mean_class_extent": extent, "extent_over_vcore": extent / vc}
    for prec in ("f32", "f32x2"):
        eng.wake_clear()
        eng.wake_append(x[order], z[order], g[order])
        u, w = eng.wake_advect(2.0 ** -10, [], [], [], vc, precision=prec, return_velocity=True)
        rec[f"err_{prec}"] = float(max(np.abs(u[slot[sel]] - ur).max(), np.abs(w[slot[sel]] - wr).max()) / scale)
    rec["err_f32_per_extent_over_vcore"] = rec["err_f32"] / rec["extent_over_vcore"]
    print(json.dumps(rec), flush=True)
