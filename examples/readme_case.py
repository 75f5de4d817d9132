#!/usr/bin/env python3
"""The reference's README example (LUDVM.py:161-162) on an MI355X, then a flow-field snapshot.

    python examples/readme_case.py [--precision auto|f32|f32x2|f64] [--plot out.png]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="auto", choices=["auto", "f32", "f32x2", "f64"])
ap.add_argument("--plot", default=None, help="write Cl(t) and the vorticity field at the last step to this PNG")
args = ap.parse_args()

sim = LUDVM(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012",
            precision=args.precision)
print(f"precision {sim.precision}, steps {sim.nt - 1}, TEVs {sim.itev + 1}, LEVs {sim.ilev}, max|LESP| {np.abs(sim.LESP).max():.4f}")
print("Cl[-3:] =", sim.Cl[-3:], " Cd[-3:] =", sim.Cd[-3:], " Cm[-3:] =", sim.Cm[-3:])
sim.flowfield(xmin=-22, xmax=0, zmin=-3, zmax=3, dr=0.02, tsteps=[sim.nt - 1])
print("flow field", sim.u_ff.shape, "max |omega| =", float(np.abs(sim.ome_ff).max()))
if args.plot:
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(2, 1, figsize=(9, 6))
    ax[0].plot(sim.t, sim.Cl, label="Cl"); ax[0].plot(sim.t, sim.Cd, label="Cd"); ax[0].plot(sim.t, sim.Cm, label="Cm")
    ax[0].set_xlabel("t"); ax[0].legend()
    ax[1].pcolormesh(sim.x_ff, sim.z_ff, sim.ome_ff[0], cmap="RdBu_r", vmin=-20, vmax=20, shading="auto")
    ax[1].set_aspect("equal"); ax[1].set_xlabel("x"); ax[1].set_ylabel("z")
    fig.tight_layout(); fig.savefig(args.plot, dpi=120)
    print("wrote", args.plot)
