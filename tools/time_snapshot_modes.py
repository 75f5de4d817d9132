import time, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM, Engine
kw = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
e = Engine(0)
for prec in ("f64", "f32"):
    for label, opts in (("all rows recorded, marched (consecutive recorded steps share a call)", dict(march=True)), ("all rows recorded, per-step path", dict(march=False)),
                        ("no rows, marched", dict(march=True, snap=[]))):
        snap = opts.pop("snap", range(401))
        LUDVM(**kw, verbose=False, engine=e, precision=prec, history="sparse", snapshot_steps=snap, **opts)
        t0 = time.perf_counter()
        for _ in range(3):
            LUDVM(**kw, verbose=False, engine=e, precision=prec, history="sparse", snapshot_steps=snap, **opts)
        print(prec, label, round((time.perf_counter() - t0) / 3 * 1e3, 1), "ms")
