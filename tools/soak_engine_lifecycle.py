#!/usr/bin/env python3
"""Engine create / run / destroy cycles: device memory must come back (buffers, events, second stream of the march)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM, Engine  # noqa: E402

kw = dict(t0=0, tf=2, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012", verbose=False)
torch.cuda.init()
free0 = None
t0 = time.perf_counter()
for cyc in range(300):
    e = Engine(0)
    if cyc % 3 == 0:
        e.set_symmetric(8)                     # overlapped march steps: second stream and its events
    LUDVM(**kw, engine=e, precision=("f32", "f64", "f32x2")[cyc % 3], history=("full", "sparse")[cyc % 2])
    e.close()
    if cyc == 20:
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(json.dumps({"cycles": 300, "seconds": round(time.perf_counter() - t0, 1), "free_after_20_cycles_MB": free0 / 2**20,
                  "free_after_300_cycles_MB": free1 / 2**20, "leaked_MB": (free0 - free1) / 2**20}))
assert free0 - free1 < 64 * 2**20, "device memory is not coming back"
