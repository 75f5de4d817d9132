#!/usr/bin/env python3
"""Symmetric kernel alone (HIP events on its stream) over a growing share of the I tiles of one wake: the slope is the
rate the chip sustains, the intercept what a launch pays once (ramp, tail).  Run on the GPU box.
    python tools/sym_tile_scaling.py [n ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LUDVM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ludvm_amd", "csrc", "libludvm_hip_exp.so"))  # measurement build: forced variants / A-B switches
from ludvm_amd import Engine  # noqa: E402
from ludvm_amd._ffi import SYM_TILE  # noqa: E402

eng = Engine(0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
if os.environ.get("SYM_T"):
    eng.set_sym_tuning(int(os.environ["SYM_T"]), int(os.environ.get("SYM_R", "0")))
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [40960, 65536, 131072, 262144]:
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    acc = torch.zeros([2 * n + 1], dtype=torch.int64, device=dev)
    scale = torch.zeros([32], dtype=torch.uint8, device=dev)
    eng.sym_scale_dev(g.data_ptr(), n, 0.065, scale.data_ptr())
    nt = (n + SYM_TILE - 1) // SYM_TILE
    rec = {"n": n, "tiles": nt, "us_by_tile_count": {}}
    for frac in (0.125, 0.25, 0.5, 0.75, 1.0):
        cnt = max(1, int(round(nt * frac)))
        reps = max(5, min(100, int(2e10 / (n * n * frac))))
        base = acc.data_ptr()

        def run():
            eng.sym_accumulate_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, cnt, 0.065, scale.data_ptr(), base,
                                   base + 8 * n, base + 16 * n)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        eng.kernel_timing(True)
        eng.kernel_time_ms(reset=True)
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ms, launches = eng.kernel_time_ms(reset=True)
        eng.kernel_timing(False)
        rec["us_by_tile_count"][cnt] = round(ms * 1e3, 1)
    c = np.array(list(rec["us_by_tile_count"].keys()), float)
    t = np.array(list(rec["us_by_tile_count"].values()), float)
    slope, icpt = np.polyfit(c[1:], t[1:], 1)
    rec["fit_us_per_tile"] = round(float(slope), 3)
    rec["fit_intercept_us"] = round(float(icpt), 1)
    rec["slope_rate_pairs_per_s"] = float(n) * SYM_TILE / (slope * 1e-6)      # a tile row = n x 512 ordered pairs
    print(json.dumps(rec), flush=True)
