"""README case, then LUDVM.flowfield on the reference's default grid (500 x 400, LUDVM.py:1186) at three time steps:
wall time of the flowfield call (velocity + vorticity).  Run on the GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ludvm_amd import LUDVM  # noqa: E402

sim = LUDVM(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012", verbose=False)
sim.flowfield(tsteps=[100])          # warm-up (allocations)
t0 = time.perf_counter()
for _ in range(5):
    sim.flowfield(tsteps=[100, 200, 300])
el = (time.perf_counter() - t0) / 5
print(json.dumps({"flowfield_default_grid_3_steps_s": el, "per_step_ms": el / 3 * 1e3, "grid": list(sim.u_ff.shape)}))
