#!/usr/bin/env python3
"""Parity at BASELINE config 2's parameters beyond the GPU tests' 300 steps: the first `--steps` time steps
(dt = 1e-3) of the CPU restatement of the reference (oracle, float64, ~45 s for 2000 steps) against the drop-in
class on the GPU, marched, in the three precisions.  Reports the largest differences per window of steps.

    python tools/parity_cfg2_2000.py --steps 2000
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ludvm_oracle as O  # noqa: E402
from ludvm_amd import LUDVM, Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    a = ap.parse_args()
    kw = dict(t0=0, tf=a.steps * 1e-3, dt=1e-3, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
    t0 = time.perf_counter()
    ref = O.OracleLUDVM(**kw)
    t_ref = time.perf_counter() - t0
    print(json.dumps({"oracle_steps": ref.nt - 1, "oracle_wall_s": t_ref, "lev_shed": int((ref.LEV_shed != -1).sum())}), flush=True)
    eng = Engine(0)
    nt = ref.nt
    edges = [1, 300, 600, 1000, 1500, nt]
    for prec in ("f64", "f32x2", "f32"):
        for march in (True, False):
            t0 = time.perf_counter()
            sim = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="sparse", march=march)
            el = time.perf_counter() - t0
            same = sim.LEV_shed == ref.LEV_shed
            first_diff = int(np.argmin(same)) if not same.all() else -1
            out = {"precision": prec, "path": "march" if march else "per-step", "wall_s": round(el, 3),
                   "lev_pattern_identical": bool(same.all()), "first_step_with_different_shedding": first_diff}
            for lo, hi in zip(edges[:-1], edges[1:]):
                if lo >= nt:
                    break
                hi = min(hi, nt)
                out[f"max_dCl_{lo}_{hi}"] = float("%.3e" % np.abs(sim.Cl[lo:hi] - ref.Cl[lo:hi]).max())
            k = min(sim.itev + 1, ref.path["TEV"].shape[2])
            out["max_dGammaTEV"] = float("%.3e" % np.abs(sim.circulation["TEV"] - ref.circulation["TEV"]).max())
            out["max_TEV_position_diff_last_step"] = float("%.3e" % np.abs(sim.path["TEV"][nt - 1][:, :k] - ref.path["TEV"][-1][:, :k]).max())
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
