#!/usr/bin/env python3
"""CPU baseline for BASELINE config 2: the reference cannot run it (two [50001, 2, 50000] float64
history arrays = 80 GB, ~1e14 pair evaluations), so time the first `--steps` time steps of the CPU
restatement of the reference's time_loop (oracle/ludvm_oracle.py: float64 NumPy broadcast, one core, dense
history as the reference keeps it) at config 2's parameters and extrapolate with the pair count of the full run.

    python tools/cfg2_cpu_baseline.py --steps 2000

Not part of the product: the oracle is test infrastructure; this script only reports a baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ludvm_oracle as O  # noqa: E402


def rollup_pairs(lev_shed_flags):
    """sum_i (n_i + 80) * n_i with n_i the wake size at step i (FREE + TEV + LEV), as tools/run_configs.py counts."""
    sizes = 1 + np.arange(1, len(lev_shed_flags) + 1) + np.cumsum(lev_shed_flags)
    return float(np.sum((sizes + 80.0) * sizes)), int(sizes[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--full-pairs", type=float, default=7.5e13, help="roll-up pairs of the full 50000-step run "
                    "(profiles/r01_config2_timeloop_march.json)")
    a = ap.parse_args()
    dt = 1e-3
    t0 = time.perf_counter()
    sim = O.OracleLUDVM(t0=0, tf=a.steps * dt, dt=dt, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2,
                        Naca="0012", verbose=False)
    el = time.perf_counter() - t0
    assert sim.nt == a.steps + 1, sim.nt
    pairs, n_last = rollup_pairs(sim.LEV_shed[1:] != -1)
    rate = pairs / el
    print(json.dumps({
        "config": f"cfg2 parameters (dt=1e-3), first {a.steps} steps, CPU restatement of the reference time_loop",
        "cores": 1, "host_logical_cpus": os.cpu_count(), "wall_s": el, "wake_after": n_last, "rollup_pairs": pairs,
        "pairs_per_s_wall": rate,
        "extrapolated_full_run_s": a.full_pairs / rate,
        "extrapolated_full_run_days": a.full_pairs / rate / 86400.0,
        "note": "extrapolation = full-run roll-up pairs / measured rate; optimistic for the CPU (its [Np, Nw] float64 "
                "temporaries fall out of cache as the wake grows, and the 80 GB of dense history is not allocated here)"}))


if __name__ == "__main__":
    main()
