#!/usr/bin/env python3
"""One simulation on several GPUs of a node in ONE process, no launcher (ludvm_amd/multi.py):

    python examples/multi_gpu_case.py --devices 8 [--tf 50 --dt 1e-3]

`LUDVM(..., devices=G)` starts a host thread, an engine and a replica of the simulation per device and joins the engines by the
library's own RCCL communicator (ncclCommInitAll); the roll-up's unordered pairs are evaluated in tile blocks with ONE integer
all-reduce per time step from `LUDVM_MIN_WAKE` vortices on, flow-field rows and large induced_velocity calls in blocks.  The
results equal the one-GPU run bit for bit (integer sums commute).  With --devices 1 this is the ordinary single-GPU run."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--devices", type=int, default=1)
ap.add_argument("--tf", type=float, default=20.0)
ap.add_argument("--dt", type=float, default=5e-2)
args = ap.parse_args()

t0 = time.perf_counter()
sim = LUDVM(t0=0, tf=args.tf, dt=args.dt, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012",
            verbose=False, history="sparse", snapshot_steps=LUDVM.flowfield_rows_needed([int(args.tf / args.dt) - 1]),
            devices=args.devices)
print(f"{args.devices} device(s): {sim.nt - 1} steps in {time.perf_counter() - t0:.2f} s, precision {sim.precision}, "
      f"TEVs {sim.itev + 1}, LEVs {sim.ilev}, Cl[-1] = {sim.Cl[-1]:.6f}")
sim.flowfield(xmin=-args.tf - 2, xmax=0, zmin=-3, zmax=3, dr=0.02, tsteps=[sim.nt - 2])      # grid rows in blocks over the devices
print("flow field", sim.u_ff.shape, "max |omega| =", float(np.abs(sim.ome_ff).max()))
if args.devices > 1:
    sim.close()          # leaves the communicator, ends the rank threads; the results stay readable
