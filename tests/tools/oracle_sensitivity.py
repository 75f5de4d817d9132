#!/usr/bin/env python3
"""How far can two equally exact float64 evaluations of the reference's time loop stay together at BASELINE config 2's
parameters (dt = 1e-3)?  Runs the CPU restatement twice: as written, and with the sources of every pair sum visited in
reverse order (same arithmetic, different rounding of the sums: a perturbation of ~1e-16 relative).  CPU only.

    python tools/oracle_sensitivity.py --steps 1500
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ludvm_oracle as O  # noqa: E402


class Reversed(O.OracleLUDVM):
    def induced_velocity(self, circulation, xw, zw, xp, zp, viscous=True):
        g, xw, zw = (np.asarray(a)[::-1] for a in (circulation, xw, zw))
        return O.induced_velocity(g, xw, zw, xp, zp, self.v_core, viscous)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    a = ap.parse_args()
    kw = dict(t0=0, tf=a.steps * 1e-3, dt=1e-3, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
    t0 = time.perf_counter()
    ref = O.OracleLUDVM(**kw)
    print(json.dumps({"run": "as written", "wall_s": round(time.perf_counter() - t0, 1),
                      "first_lev_step": int(np.argmax(ref.LEV_shed != -1)), "lev": int((ref.LEV_shed != -1).sum())}), flush=True)
    alt = Reversed(**kw)
    nt = ref.nt
    same = ref.LEV_shed == alt.LEV_shed
    out = {"run": "sources summed in reverse order vs as written", "lev_pattern_identical": bool(same.all()),
           "first_step_with_different_shedding": int(np.argmin(same)) if not same.all() else -1}
    edges = [1, 300, 600, 800, 1000, 1200, 1500, 2001]
    for lo, hi in zip(edges[:-1], edges[1:]):
        if lo >= nt:
            break
        hi = min(hi, nt)
        out[f"max_dCl_{lo}_{hi}"] = float("%.3e" % np.abs(ref.Cl[lo:hi] - alt.Cl[lo:hi]).max())
    out["max_abs_Cl"] = float("%.3e" % np.abs(ref.Cl[1:]).max())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
