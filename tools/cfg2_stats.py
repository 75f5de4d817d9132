#!/usr/bin/env python3
"""Whole-horizon statistics of BASELINE config 2 (NACA0012 sinusoidal pitch, dt = 1e-3, t in [0, 50]: 50 000 steps of
LUDVM.time_loop, reference LUDVM.py:597-1171) -- the statistical side of the parity contract.

The flow is chaotic: two float64 evaluations of the reference's own scheme that differ only in the order of a sum
decorrelate after ~1 450 steps, so beyond the first ~1 000 steps "equal to the reference" can only
mean "statistically indistinguishable from a float64 run".  This tool produces

  * the statistical reference: the full run with every pair sum in float64 on the GPU (`--runs f64`), and a second
    float64 run whose sums are split differently (`f64b`: ludvm_set_tuning source_splits = 7) -- their distance is the
    chaos band, i.e. what an equally exact evaluation can differ by;
  * the same statistics for the fp32 modes (`f32`, `f32x2`, marched or per-step),

and prints / stores, per run: per-period mean and rms of Cl, Cd, Cm (period = 1 / f = 10 time units = 10 000 steps),
LEV and TEV counts, the Kelvin residual, max |LESP|, wall time.  tests/test_gpu_cfg2_stats.py compares a run against the
committed float64 statistics (tests/golden/cfg2_f64_stats.json) with tolerances derived from the band.

    python tools/cfg2_stats.py --runs f64,f64b,f32,f32x2 --out gpurun_out/cfg2_stats.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG2 = dict(t0=0, tf=50, dt=1e-3, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")


def statistics(sim, periods=5):
    """Size-independent observables of a finished run (all float64 host arrays of the LUDVM object)."""
    nt = sim.nt
    per = (nt - 1) // periods
    out = {"steps": nt - 1, "period_steps": per}
    for name in ("Cl", "Cd", "Cm"):
        a = getattr(sim, name)
        out[name + "_mean"] = [float(a[1 + p * per:1 + (p + 1) * per].mean()) for p in range(periods)]
        out[name + "_rms"] = [float(a[1 + p * per:1 + (p + 1) * per].std()) for p in range(periods)]
        # robust spread: median absolute deviation (the rms is dominated by a few load spikes of |Cl| ~ 70 per period)
        out[name + "_mad"] = [float(np.median(np.abs(a[1 + p * per:1 + (p + 1) * per] - np.median(a[1 + p * per:1 + (p + 1) * per]))))
                              for p in range(periods)]
        out[name + "_absmax"] = float(np.abs(a[1:]).max())
    C = sim.circulation
    shed = sim.LEV_shed[1:] != -1
    n_lev = int(shed.sum())
    out["lev"], out["tev"] = n_lev, int(sim.nt - 1)
    out["lev_per_period"] = [int(shed[p * per:(p + 1) * per].sum()) for p in range(periods)]
    out["first_lev_step"] = int(np.argmax(shed)) + 1 if n_lev else -1
    # Kelvin (LUDVM.py:698-699, :758-762): bound + sum TEV + sum LEV + sum FREE = IC after every step
    ctev = np.cumsum(C["TEV"][:nt - 1])
    lev_cum = np.concatenate([[0.0], np.cumsum(C["LEV"][:n_lev])])
    clev = lev_cum[np.cumsum(shed)]
    kel = C["bound"][:nt - 1] + ctev + clev + float(np.sum(C["FREE"])) - float(C["IC"])
    out["kelvin_residual_max"] = float(np.abs(kel).max())
    out["gamma_abs_sum"] = float(np.abs(C["TEV"][:nt - 1]).sum() + np.abs(C["LEV"][:n_lev]).sum())
    out["max_abs_LESP"] = float(np.abs(sim.LESP[:nt - 1]).max())
    out["Cl_last"] = float(sim.Cl[-1])
    return out


def run_one(kind, args):
    from ludvm_amd import LUDVM, Engine
    eng = Engine(0)
    base = kind.split(":")[0]
    prec = "f64" if base.startswith("f64") else base
    if base == "f64b":
        eng.set_tuning(0, 7)           # same arithmetic, sums cut into different partial sums
    elif base.startswith("f64s"):
        eng.set_tuning(0, int(base[4:]))   # f64s<k>: every launch's sources cut into k partial sums
    march = ":step" not in kind
    kw = dict(CFG2)
    kw["tf"] = args.tf
    t0 = time.perf_counter()
    sim = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="sparse", snapshot_steps=[], march=march)
    wall = time.perf_counter() - t0
    st = statistics(sim, periods=max(1, int(round(args.tf / 10))))
    st.update(run=kind, precision=sim.precision, march=march, wall_s=wall)
    series = np.stack([sim.Cl, sim.Cd, sim.Cm]).astype(np.float64)
    shed = (sim.LEV_shed != -1)
    eng.close()
    return st, series, shed


def compare(a, b):
    """Largest per-period differences between two statistics records."""
    d = {}
    for name in ("Cl", "Cd", "Cm"):
        for st in ("mean", "rms", "mad"):
            d[f"d_{name}_{st}"] = float(np.max(np.abs(np.array(a[f"{name}_{st}"]) - np.array(b[f"{name}_{st}"]))))
    d["d_lev"] = abs(a["lev"] - b["lev"])
    d["d_lev_per_period"] = int(np.max(np.abs(np.array(a["lev_per_period"]) - np.array(b["lev_per_period"]))))
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", default="f64,f64b,f32,f32x2")
    ap.add_argument("--tf", type=float, default=50.0)
    ap.add_argument("--out", default="gpurun_out/cfg2_stats.json")
    ap.add_argument("--series", default="", help="also store the Cl/Cd/Cm series and shedding flags (npz)")
    ap.add_argument("--golden", default="", help="write the float64 ensemble record the GPU test compares against (json) "
                    "and, next to it, the first 1000 steps of the first run's loads (npz)")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    recs, series, sheds = {}, {}, {}
    for kind in a.runs.split(","):
        st, ser, sh = run_one(kind, a)
        recs[kind], series[kind], sheds[kind] = st, ser, sh
        print(json.dumps(st), flush=True)
        with open(a.out, "w") as f:
            json.dump({"runs": recs}, f, indent=1)
    kinds = list(recs)
    cmp_ = {}
    for k in kinds[1:]:
        c = compare(recs[kinds[0]], recs[k])
        both = np.nonzero(sheds[kinds[0]] != sheds[k])[0]
        c["first_step_shedding_differs"] = int(both[0]) if len(both) else -1
        dcl = np.abs(series[kinds[0]][0] - series[k][0])
        c["dCl_first_300"], c["dCl_first_1000"] = float(dcl[:301].max()), float(dcl[:1001].max())
        cmp_[f"{kinds[0]} vs {k}"] = c
        print(f"{kinds[0]} vs {k}:", json.dumps(c), flush=True)
    with open(a.out, "w") as f:
        json.dump({"runs": recs, "compare": cmp_}, f, indent=1)
    if a.golden:
        ens = [k for k in kinds if k.startswith("f64")]
        stats = {}
        for name in ("Cl_mean", "Cl_rms", "Cl_mad", "Cd_mean", "Cd_rms", "Cd_mad", "Cm_mean", "Cm_rms", "Cm_mad", "lev_per_period"):
            v = np.array([recs[k][name] for k in ens], dtype=float)          # [runs, periods]
            var = v.var(0, ddof=1)
            # periods 2.. are statistically alike (the starting transient is over): their variances pooled give a
            # spread estimate with (runs - 1) * (periods - 1) degrees of freedom instead of runs - 1
            stats[name] = {"mean": v.mean(0).tolist(), "std": np.sqrt(var).tolist(), "std_pooled": float(np.sqrt(var[1:].mean())),
                           "min": v.min(0).tolist(), "max": v.max(0).tolist()}
        for name in ("lev", "Cl_absmax", "Cd_absmax", "Cm_absmax", "gamma_abs_sum"):
            v = np.array([recs[k][name] for k in ens], dtype=float)
            stats[name] = {"mean": float(v.mean()), "std": float(v.std(ddof=1)), "min": float(v.min()), "max": float(v.max())}
        first = recs[ens[0]]
        gold = {"what": "BASELINE config 2 (dt = 1e-3, t in [0, 50]), float64 pair sums on the GPU, ensemble over the "
                        "partition of every sum into partial sums: the spread is what equally exact evaluations differ by",
                "generated_by": "tools/cfg2_stats.py --golden (MI355X)", "runs": ens, "n_runs": len(ens),
                "period_steps": first["period_steps"], "steps": first["steps"], "first_lev_step": first["first_lev_step"],
                "tev": first["tev"], "max_abs_LESP": max(recs[k]["max_abs_LESP"] for k in ens),
                "kelvin_residual_max": max(recs[k]["kelvin_residual_max"] for k in ens), "ensemble": stats,
                "per_run": {k: recs[k] for k in ens}}
        with open(a.golden, "w") as f:
            json.dump(gold, f, indent=1)
        np.savez_compressed(os.path.splitext(a.golden)[0] + "_first1000.npz", loads=series[ens[0]][:, :1001],
                            shed=sheds[ens[0]][:1001])
    if a.series:
        np.savez_compressed(a.series, **{k.replace(":", "_"): series[k].astype(np.float32) for k in kinds})
