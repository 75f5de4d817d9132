#!/usr/bin/env python3
"""A/B of the device-resident march against the per-step path (same engine, same inputs).

    python tools/march_check.py [--tf 20] [--dt 5e-2] [--precision f64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM  # noqa: E402


ENG = None


def run(march, a):
    global ENG
    if ENG is None:
        from ludvm_amd import Engine
        ENG = Engine(0)
        if getattr(a, "sym_threshold", 0):
            ENG.set_symmetric(a.sym_threshold)     # symmetric kernel (and overlapped march steps) from this wake size
    t0 = time.perf_counter()
    s = LUDVM(t0=0, tf=a.tf, dt=a.dt, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca='0012',
              verbose=False, precision=a.precision, history='sparse', march=march, engine=ENG)
    return s, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tf", type=float, default=20.0)
    ap.add_argument("--dt", type=float, default=5e-2)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--sym-threshold", type=int, default=0, help="wake size from which the symmetric kernel is used "
                    "(0 = library default 16384); small values exercise the overlapped march steps on short runs")
    a = ap.parse_args()
    run(True, argparse.Namespace(tf=1.0, dt=a.dt, precision=a.precision, sym_threshold=a.sym_threshold))   # warm-up (library load, allocations)
    sm, tm = run(True, a)
    sp, tp = run(False, a)
    nt = sm.nt
    out = {"nt": nt, "precision": a.precision, "sym_threshold": a.sym_threshold, "wall_march_s": tm, "wall_per_step_s": tp,
           "lev_pattern_identical": bool(np.array_equal(sm.LEV_shed != -1, sp.LEV_shed != -1)),
           "n_lev": int((sm.LEV_shed != -1).sum())}
    for lo, hi in ((1, 50), (50, 100), (100, 200), (200, nt)):
        hi = min(hi, nt)
        if lo >= hi:
            continue
        out[f"max_dCl_{lo}_{hi}"] = float(np.abs(sm.Cl[lo:hi] - sp.Cl[lo:hi]).max())
    out["max_dGammaTEV_first100"] = float(np.abs(sm.circulation['TEV'][:100] - sp.circulation['TEV'][:100]).max())
    out["max_dfourier_first100"] = float(np.abs(sm.fourier[:100] - sp.fourier[:100]).max())
    out["max_dbound_first100"] = float(np.abs(sm.circulation['bound'][:100] - sp.circulation['bound'][:100]).max())
    out["max_dCm_first100"] = float(np.abs(sm.Cm[1:100] - sp.Cm[1:100]).max())
    out["max_dCd_first100"] = float(np.abs(sm.Cd[1:100] - sp.Cd[1:100]).max())
    out["max_dgamma_airfoil_first100"] = float(np.abs(sm.circulation['gamma_airfoil'][:100]
                                                     - sp.circulation['gamma_airfoil'][:100]).max())
    out["last_row_TEV_maxdiff"] = float(np.abs(sm.path['TEV'][nt - 1] - sp.path['TEV'][nt - 1]).max())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
