#!/usr/bin/env python3
"""Measured distance of the fp32 time loops from the reference's golden README run (tests/golden/g2_config1.npz) in the
windows the T2 tests bound: max |dCl|, |dCd|, |dCm| over steps < 50 / 75 / 100 / 150, per precision and path.
    python tools/t2_windows.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine  # noqa: E402

CONFIG1 = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))


def windows(sim):
    out = {}
    for hi in (50, 75, 100, 150):
        out[hi] = max(float(np.abs(getattr(sim, n)[:hi] - g2[n][:hi]).max()) for n in ("Cl", "Cd", "Cm"))
    return out


for thr in (1, 8):
    eng = Engine(0)
    eng.set_symmetric(thr)
    for prec in ("f32", "f32x2"):
        for march in (True, False):
            sim = LUDVM(**CONFIG1, verbose=False, engine=eng, precision=prec, history="sparse", march=march)
            print(json.dumps({"sym_threshold": thr, "precision": prec, "march": march, "same_shedding": bool(np.array_equal(sim.LEV_shed, g2["LEV_shed"])),
                              "max_abs_load_diff_before_step": windows(sim)}), flush=True)
    eng.close()
