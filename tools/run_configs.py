#!/usr/bin/env python3
"""Measure the BASELINE configs that are not bench.py's headline line (run on the GPU box).

    python tools/run_configs.py cfg5            # flow field 4096 x 4096 grid over N = 1e6 vortices
    python tools/run_configs.py cfg2 --tf 50    # NACA0012 sinusoidal pitch, dt = 1e-3, full time_loop
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cfg5(args):
    import torch
    from ludvm_amd import Engine
    n, nx, nz = args.vortices, args.grid, args.grid
    rng = np.random.default_rng(20260101)
    x = rng.uniform(-10, 0, n).astype(np.float32)
    z = rng.uniform(-2, 2, n).astype(np.float32)
    g = (rng.standard_normal(n) / n).astype(np.float32)
    xmin, zmin, dr = -8.0, -4.0, 8.0 / nx
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_tuning(args.tpl, args.splits)      # tpl != 0 forces the generic kernel (no shared-dx grid variant); splits: source splits
    dx, dz, dg = (torch.from_numpy(a).to(dev) for a in (x, z, g))
    du = torch.empty(nx * nz, dtype=torch.float32, device=dev)
    dw = torch.empty_like(du)
    dome = torch.empty_like(du)

    def run():
        eng.flowfield_dev(xmin, zmin, dr, nx, nz, dx.data_ptr(), dz.data_ptr(), dg.data_ptr(), n, 0.065,
                          du.data_ptr(), dw.data_ptr())
        eng.vorticity_dev(du.data_ptr(), dw.data_ptr(), nx, nz, dr, dome.data_ptr())
    run()
    torch.cuda.synchronize()
    eng.kernel_timing(True)
    eng.kernel_time_ms(True)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.reps
    kms, _ = eng.kernel_time_ms(True)
    pairs = float(nx) * nz * n
    # (parity of this case on sampled grid points against the C oracle: tests/test_gpu_kernel.py::test_full_size_config5_flowfield)
    err = None
    print(json.dumps({"config": f"cfg5 flowfield {nx}x{nz} grid over N={n}", "kernel": "generic tpl=%d" % args.tpl if args.tpl else {"row": "grid row of 4 (shared dx)", "1": "grid row of 4 (shared dx)", "patch2": "grid patch 2 x 4 (shared dx, dz, G dx, G dz)"}.get(os.environ.get("LUDVM_GRID_KERNEL", ""), "grid patch 4 x 4 (shared dx, dz, G dx, G dz)"), "s_per_call": el, "pairs_per_s": pairs / el,
                      "pair_kernel_ms": kms, "pct_fp32_peak": 13 * pairs / (kms * 1e-3) / 157.3e12 * 100,
                      "sampled_rel_err_vs_oracle": err, "omega_finite": bool(torch.isfinite(dome).all().item())}))


def cfg2(args):
    from ludvm_amd import LUDVM, Engine
    eng = Engine(0)
    if args.sym_threshold:
        eng.set_symmetric(args.sym_threshold)
    eng.kernel_timing(not args.no_timing)    # per-launch events (two per pair-kernel launch)
    t0 = time.perf_counter()
    sim = LUDVM(t0=0, tf=args.tf, dt=args.dt, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2,
                Naca="0012", verbose=args.verbose, engine=eng, precision=args.precision, history="sparse",
                snapshot_steps=[], march=not args.no_march, run=False)
    t_setup = time.perf_counter() - t0          # geometry + kinematics of all steps (host)
    sim.time_loop()
    sim.compute_coefficients()
    el = time.perf_counter() - t0
    kms, nl = eng.kernel_time_ms(True)
    ntev, nlev = sim.itev + 1, sim.ilev + (1 if sim.LEV_shed[-1] != -1 else 0)
    # pairs of the roll-up launches: sum_i (n_i + 80) * n_i with n_i = wake size at step i
    sizes = 1 + np.arange(1, sim.nt) + np.cumsum(sim.LEV_shed[1:] != -1)
    pairs = float(np.sum((sizes + 80.0) * sizes))
    print(json.dumps({"config": f"cfg2 time_loop dt={args.dt:g} tf={args.tf} precision={args.precision}", "steps": sim.nt - 1,
                      "path": "per-step round trips" if args.no_march else "device-resident march",
                      "wall_s": el, "setup_s": t_setup, "time_loop_s": el - t_setup, "final_wake": int(sizes[-1]), "tev": int(ntev), "lev": int(nlev),
                      "rollup_pairs": pairs, "pairs_per_s_wall": pairs / el, "kernel_launches": nl,
                      "kernel_ms_total": kms * nl,
                      # the wake size after every 250th step (the wake only grows: tools/roofline_table.py turns the call
                      # counts of a kernel-statistics file into step ranges and mean pairs per launch with it)
                      "wake_size_every_250_steps": [int(v) for v in sizes[::250]],
                      "Cl_last": float(sim.Cl[-1]), "Cl_mean_last_period": float(np.mean(sim.Cl[-10000:])),
                      "max_abs_LESP": float(np.abs(sim.LESP).max())}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["cfg5", "cfg2"])
    ap.add_argument("--vortices", type=int, default=1_000_000)
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--tpl", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0, help="cfg5: source splits (0 = the library's rule)")
    ap.add_argument("--tf", type=float, default=50.0)
    ap.add_argument("--dt", type=float, default=1e-3, help="cfg2: time step (BASELINE config 2 uses 1e-3)")
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--no-march", action="store_true", help="cfg2: one device round trip per time step")
    ap.add_argument("--sym-threshold", type=int, default=0, help="cfg2: smallest wake that takes the symmetric kernel (0 = library default)")
    ap.add_argument("--no-timing", action="store_true", help="cfg2: no per-launch timing events (kernel_ms_total reads 0)")
    a = ap.parse_args()
    if os.environ.get("LUDVM_PROFILE_PMC") == "1":      # counter passes (tools/profile_cmd.sh): one repetition is enough
        a.reps = 1
    {"cfg5": cfg5, "cfg2": cfg2}[a.which](a)
