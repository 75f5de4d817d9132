"""cProfile of the README case (config 1, dense history) on the drop-in class (run on the GPU box)."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM, Engine  # noqa: E402

eng = Engine(0)
kw = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012", verbose=False,
          engine=eng)
LUDVM(**dict(kw, tf=1))
for prec in ("f32", "f64"):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        LUDVM(**kw, precision=prec)
        best = min(best, time.perf_counter() - t0)
    print(f"config 1, dense history, {prec}: {best * 1e3:.1f} ms per run (400 steps)")
    t0 = time.perf_counter()
    LUDVM(**kw, precision=prec, history="sparse")
    print(f"config 1, sparse history (march), {prec}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
LUDVM(**kw, precision="f32")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
