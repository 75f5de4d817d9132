#!/usr/bin/env python3
"""Timeline of a MARCHED time step (ludvm_march_run) at given wake sizes: how long the side chain (chord sums -> solve) is
next to the roll-up kernel it runs beside (overlapped steps) or in front of (serial steps).  Two halves:

    python tools/march_timeline.py run <nf>            # the workload: a sheet of nf weak free vortices, spaced like config 2's
                                                       # wake (1e-3 chords apart, stored along itself), STEPS marched steps
    python tools/march_timeline.py read <trace.csv> <nf>   # per-step medians from a rocprofv3 --kernel-trace of the above

tools/march_timeline.sh runs both per size (GPU box).  Reported per size, medians over the second half of the steps:
  period        start of a step's roll-up kernel -> start of the next step's
  rollup        duration of the pair kernel (pair_sym_* or pair_f32<...> of the roll-up)
  chain         start of the chord-sum kernel -> end of march_solve (the critical path beside / before the roll-up)
  per kernel    duration and the gap in front of it on its own queue
"""
import csv
import json
import os
import statistics as st
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(nf):
    import numpy as np
    from ludvm_amd import LUDVM, Engine
    steps = int(os.environ.get("STEPS", "300"))
    dt = 1e-3
    eng = Engine(0)
    rng = np.random.default_rng(5)
    s = np.linspace(0.0, 1.0, nf)
    # config 2's spacing: one vortex per Uinf dt = 1e-3 chords, a gently waving sheet behind the foil
    xy = np.stack([2.0 + 1e-3 * nf * s, 0.05 * np.sin(6.0 * 1e-3 * nf * s)])
    gam = rng.standard_normal(nf) * 1e-4
    kw = dict(t0=0, tf=(steps - 0.5) * dt, dt=dt, verbose=False, engine=eng, precision="f32", history="sparse",
              circulation_freevort=gam, xy_freevort=xy)
    LUDVM(**dict(kw, tf=63.5 * dt))
    t0 = time.perf_counter()
    sim = LUDVM(**kw)
    wall = time.perf_counter() - t0
    print(json.dumps({"free_vortices": nf, "wake_at_end": int(eng.wake_size()), "steps": sim.nt - 1,
                      "us_per_step": round(wall / (sim.nt - 1) * 1e6, 1), "Cl_last": float(sim.Cl[-1])}), flush=True)


def read(path, nf):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        r["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ludvm::", "")
        r["t0"], r["t1"] = int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3

    def is_rollup(n):
        return n.startswith("pair_sym") or (n.startswith("pair_f32") and "true>" in n)
    def is_chord(n):
        return n.startswith("pair_f64")
    solves = [r for r in rows if r["name"].startswith("march_solve")]
    solves = solves[len(solves) // 2:]                # the last run's second half: steady state
    if len(solves) < 8:
        print(f"n={nf}: too few marched steps in the trace")
        return
    t_lo = solves[0]["t0"] - 2000.0
    rows = [r for r in rows if r["t0"] >= t_lo]
    per = {}
    chains, rollups, periods, exposed = [], [], [], []
    last_roll = None
    prev_end_by_queue = {}
    chord = None
    for r in rows:
        q = r.get("Queue_Id", "0")
        gap = r["t0"] - prev_end_by_queue.get(q, r["t0"])
        prev_end_by_queue[q] = r["t1"]
        per.setdefault(r["name"], []).append((gap, r["t1"] - r["t0"]))
        if is_chord(r["name"]):
            chord = r
        elif r["name"].startswith("march_solve") and chord is not None:
            chains.append(r["t1"] - chord["t0"])
            chord = None
        elif is_rollup(r["name"]):
            # (the quad variant's diagonal launch precedes it: count the longer one)
            if last_roll is not None and r["t0"] - last_roll["t0"] > 1.0:
                periods.append(r["t0"] - last_roll["t0"])
            rollups.append(r["t1"] - r["t0"])
            last_roll = r
    med = lambda v: st.median(v) if v else float("nan")      # noqa: E731
    queues = len({r.get("Queue_Id", "0") for r in rows})
    print(f"n={nf}: period {med(periods):7.1f} us   rollup {med(rollups):7.1f} us   chain (chord start -> solve end) {med(chains):6.1f} us"
          f"   [{len(chains)} steps, {queues} queue(s): {'overlapped' if queues > 1 else 'serial'}; under the profiler]")
    for name, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        if len(v) < len(chains) // 2:
            continue
        print(f"     {name[:70]:70s} x{len(v):5d}  gap before {med([g for g, _ in v]):6.1f} us   duration {med([d for _, d in v]):7.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        read(sys.argv[2], int(sys.argv[3]))
