#!/usr/bin/env python3
"""When do the waves of one symmetric-kernel launch start and end?  Needs the measurement build of the library
(hipcc ... -DLUDVM_WAVE_TRACE, loaded through LUDVM_HIP_LIB): every wave stores two 100 MHz time stamps.
    LUDVM_HIP_LIB=/path/to/lib_trace.so python tools/sym_wave_trace.py [n ...]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import Engine  # noqa: E402
from ludvm_amd._ffi import SYM_TILE  # noqa: E402

eng = Engine(0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
lib = eng._lib
lib.ludvm_debug_set_wave_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.ludvm_debug_set_wave_trace.restype = ctypes.c_int
rng = np.random.default_rng(3)
if os.environ.get("SYM_R"):
    eng.set_sym_tuning(8, int(os.environ["SYM_R"]))
for n in [int(a) for a in sys.argv[1:]] or [40960, 65536, 131072]:
    x = torch.from_numpy(rng.uniform(-10, 0, n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(-2, 2, n).astype(np.float32)).to(dev)
    g = torch.from_numpy((rng.standard_normal(n) / n).astype(np.float32)).to(dev)
    acc = torch.zeros([2 * n + 1], dtype=torch.int64, device=dev)
    scale = torch.zeros([32], dtype=torch.uint8, device=dev)
    eng.sym_scale_dev(g.data_ptr(), n, 0.065, scale.data_ptr())
    nt = (n + SYM_TILE - 1) // SYM_TILE
    trace = torch.zeros([1 << 21], dtype=torch.int64, device=dev)
    base = acc.data_ptr()

    def run():
        eng.sym_accumulate_dev(x.data_ptr(), z.data_ptr(), g.data_ptr(), n, 0, nt, 0.065, scale.data_ptr(), base, base + 8 * n,
                               base + 16 * n)
    # WARM_SECONDS of back-to-back launches first (default 0.3): the first 20-30 ms after an idle period run ~14 % slower
    import time
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < float(os.environ.get("WARM_SECONDS", "0.3")):
        for _ in range(10):
            run()
        torch.cuda.synchronize()
    assert lib.ludvm_debug_set_wave_trace(eng._ctx, ctypes.c_void_p(trace.data_ptr())) == 0
    trace.zero_()
    run()
    torch.cuda.synchronize()
    assert lib.ludvm_debug_set_wave_trace(eng._ctx, None) == 0
    t = trace.cpu().numpy().reshape(-1, 2)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0          # microseconds
    end = (t[:, 1] - t0) / 100.0
    total = end.max()
    edges = np.linspace(0, total, 41)
    running = [(int(((start <= e) & (end > e)).sum())) for e in edges[:-1]]
    life = end - start
    rec = {"n": n, "waves": int(len(t)), "kernel_us": round(float(total), 1),
           "all_started_by_us": round(float(start.max()), 1), "first_end_us": round(float(end.min()), 1),
           "lifetime_us": {"min": round(float(life.min()), 1), "median": round(float(np.median(life)), 1), "max": round(float(life.max()), 1)},
           "last_10pct_of_waves_end_after_us": round(float(np.percentile(end, 90)), 1),
           "busy_wave_time_over_kernel_time_x_peak_running": round(float(life.sum() / (total * max(running))), 3),
           "running_waves_in_40_bins": running}
    print(json.dumps(rec), flush=True)
