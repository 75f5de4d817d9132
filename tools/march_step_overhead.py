#!/usr/bin/env python3
"""What a marched time step costs beyond its symmetric pair kernel, at a given wake size (GPU box).  A SHEET of `nf` weak
free vortices behind the foil, stored in order along itself, stands for an old wake (round 5: rounds 3-4 used a random cloud,
which the order-independent fp32 tier of round 4 turned into another case -- too sparse for its core, hi+lo positions, other
kernels than a shed wake takes); `steps` steps of config 2's dt are marched (sparse history, fp32) and timed as a whole; the pair
kernel's own time comes from HIP events around it (ludvm_kernel_timing) in a second, equal run.  NOT the source of DESIGN.md's
config-2 roofline rows any more: those come from config 2's own run (tools/profile_batch.sh, tools/roofline_table.py).
    python tools/march_step_overhead.py [nf ...]        STEPS=600"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine  # noqa: E402

steps = int(os.environ.get("STEPS", "600"))
dt = 1e-3
eng = Engine(0)
for nf in [int(a) for a in sys.argv[1:]] or [20000, 32768, 49152, 65536]:
    rng = np.random.default_rng(5)
    sline = np.linspace(0.0, 1.0, nf)
    xy = np.stack([2.0 + 48.0 * sline, 0.5 * np.sin(40.0 * sline)])          # 48 chords of wavy sheet, in shedding order
    gam = rng.standard_normal(nf) * 1e-4
    kw = dict(t0=0, tf=(steps - 0.5) * dt, dt=dt, verbose=False, engine=eng, precision="f32", history="sparse",
              circulation_freevort=gam, xy_freevort=xy)
    LUDVM(**dict(kw, tf=63.5 * dt))                     # warm-up (allocations, clocks)
    t0 = time.perf_counter()
    sim = LUDVM(**kw)
    wall = time.perf_counter() - t0
    eng.kernel_timing(True)
    eng.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    sim2 = LUDVM(**kw)
    wall_timed = time.perf_counter() - t0
    ms, nl = eng.kernel_time_ms(reset=True)
    eng.kernel_timing(False)
    assert np.array_equal(sim.Cl, sim2.Cl)
    n_end = eng.wake_size()
    print(json.dumps({"free_vortices": nf, "wake_at_end": int(n_end), "steps": sim.nt - 1,
                      "us_per_step": round(wall / (sim.nt - 1) * 1e6, 1),
                      "pair_kernel_us_avg": round(ms * 1e3, 1), "pair_kernel_launches": int(nl),
                      "us_per_step_with_event_timing": round(wall_timed / (sim.nt - 1) * 1e6, 1)}), flush=True)
