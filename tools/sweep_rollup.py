"""One roll-up step of the resident wake (ludvm_wake_advect: pair kernel + Euler finisher, no host read-back) over the
wake size, per precision and kernel: microseconds per step.  Run on the GPU box.
    python tools/sweep_rollup.py [sizes...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LUDVM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ludvm_amd", "csrc", "libludvm_hip_exp.so"))  # measurement build: forced variants / A-B switches
from ludvm_amd import Engine  # noqa: E402

eng = Engine(0)
if os.environ.get("SYM_T"):
    eng.set_sym_tuning(int(os.environ["SYM_T"]), int(os.environ.get("SYM_R", "0")))
if os.environ.get("SYM_YS"):
    eng.set_tuning(0, int(os.environ["SYM_YS"]))       # d-chunks per tile of the symmetric kernel (and source splits of the direct one)
rng = np.random.default_rng(1)
sizes = [int(a) for a in sys.argv[1:]] or [8192, 11264, 12288, 14336, 16384, 20480, 24576, 32768, 40960, 49152, 65536]
fx, fz, fg = np.linspace(-30.9, -30.0, 80), np.zeros(80), rng.standard_normal(80) / 100
for n in sizes:
    x = -30.0 + np.sort(rng.uniform(0, 1e-3 * n, n))
    z = 0.3 * np.sin(0.7 * x) + 1e-3 * rng.standard_normal(n)
    if os.environ.get("SWEEP_CLOUD"):          # a random cloud instead of a sheet (same arithmetic, other operand bits)
        x, z = rng.uniform(2.0, 12.0, n), rng.uniform(-1.5, 1.5, n)
    g = rng.standard_normal(n) * 1e-3
    res = {"n": n}
    for mode, mname in ((2, "sym"),) + (() if os.environ.get("SWEEP_SYM_ONLY") else ((0, "direct"),)):
        eng.set_symmetric(mode)
        for prec in (("f32",) if os.environ.get("SWEEP_F32_ONLY") else ("f32", "f32x2")):
            eng.wake_clear()
            eng.wake_append(x, z, g)
            # SWEEP_SECONDS (default 0.3) of back-to-back steps per timed block, three blocks, the median.  SWEEP_SECONDS=0 gives
            # the blocks rounds 1-3 used (>= 5 steps: ~10 ms at 1e5 vortices, right after the upload's idle time -- the first
            # 20-30 ms after an idle period run ~14 % slower, profiles/r03_sweep_rollup_sustained.txt)
            secs = float(os.environ.get("SWEEP_SECONDS", "0.3"))
            reps = max(5, min(300, int(4e10 / (n * n)))) if secs <= 0 else max(5, int(secs * 8.5e12 / (n * n)))
            for _ in range(3):
                eng.wake_advect(1e-6, fx, fz, fg, 1.3e-3, precision=prec)
            eng.synchronize()
            blocks = []
            for _ in range(1 if secs <= 0 else 3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.wake_advect(1e-6, fx, fz, fg, 1.3e-3, precision=prec)
                eng.synchronize()
                blocks.append((time.perf_counter() - t0) / reps * 1e6)
            res[f"{mname}_{prec}_us"] = round(sorted(blocks)[len(blocks) // 2], 1)
            if os.environ.get("SWEEP_KERNEL_TIME"):     # the pair kernel alone (HIP events around it), a further block
                eng.kernel_timing(True)
                eng.kernel_time_ms(reset=True)
                for _ in range(reps):
                    eng.wake_advect(1e-6, fx, fz, fg, 1.3e-3, precision=prec)
                eng.synchronize()
                ms, nl = eng.kernel_time_ms(reset=True)
                eng.kernel_timing(False)
                res[f"{mname}_{prec}_kernel_us"] = round(ms * 1e3, 1)
            if secs > 0:
                res[f"{mname}_{prec}_blocks_us"] = [round(b, 1) for b in blocks]
    print(json.dumps(res), flush=True)
