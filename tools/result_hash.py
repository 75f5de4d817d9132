#!/usr/bin/env python3
"""Bit-level fingerprints of the pair kernels' results, to A/B two builds of the library on one box (GPU box):
    python tools/result_hash.py [--lib path/to/libludvm_hip.so]
prints one JSON line: sha256 of the (u, w) bits of self-interaction calls at sizes that take each symmetric variant
(plain, mixed granularity, quad) and of the direct kernel, plus a flow-field patch."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--sizes", type=int, nargs="*", default=[20000, 40000, 70000, 200000, 400000, 600001, 1000000])
a = ap.parse_args()
from ludvm_amd import _ffi  # noqa: E402
if a.lib:
    _ffi.LIB_PATH = os.path.abspath(a.lib)
import torch  # noqa: E402
from ludvm_amd import Engine  # noqa: E402

eng = Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
out = {"lib": _ffi.LIB_PATH}
for n in a.sizes:
    rng = np.random.default_rng(n)
    x, z = rng.uniform(-10, 0, n).astype(np.float32), rng.uniform(-2, 2, n).astype(np.float32)
    g = (rng.standard_normal(n) / n).astype(np.float32)
    dx, dz, dg = (torch.from_numpy(v).to(dev) for v in (x, z, g))
    du, dw = torch.empty_like(dx), torch.empty_like(dx)
    for sym in (1, 0):
        if sym == 0 and n > 200000:
            continue
        eng.set_symmetric(sym)
        eng.induce_dev(dx.data_ptr(), dz.data_ptr(), dg.data_ptr(), n, dx.data_ptr(), dz.data_ptr(), n, 0.065, du.data_ptr(), dw.data_ptr())
        torch.cuda.synchronize()
        h = hashlib.sha256(du.cpu().numpy().tobytes() + dw.cpu().numpy().tobytes()).hexdigest()[:16]
        out[f"{'sym' if sym else 'direct'}_{n}"] = h
print(json.dumps(out))
