#!/usr/bin/env python3
"""Wall time of the reference's README case (LUDVM() with its default arguments: 800 steps of dt = 1.5e-2) and of config 1
(tf = 20, dt = 5e-2: 400 steps) on the drop-in class: whole constructor, best of 5 (GPU box).
    python tools/time_readme_case.py [precision ...]      (default: auto f32)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine  # noqa: E402

eng = Engine(0)
CONFIG1 = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
for prec in sys.argv[1:] or ["auto", "f32"]:
    for name, kw in (("config 1 (400 steps)", CONFIG1), ("README defaults", {})):
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            sim = LUDVM(**kw, verbose=False, engine=eng, precision=prec)
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"case": name, "precision": prec, "steps": sim.nt - 1, "wall_ms": round(best * 1e3, 2),
                          "us_per_step": round(best / (sim.nt - 1) * 1e6, 1), "wake": int(eng.wake_size())}), flush=True)
