#!/usr/bin/env python3
"""Per-step trace of one configuration: |Cl(march) - Cl(per-step)| and |Cl - Cl(f64)| for both paths, with the wake size,
to see WHERE two evaluations part (GPU box).  FUZZ_CASE_KW as in tools/fuzz_case_probe.py; optional argument: precision."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine  # noqa: E402

kw = dict(t0=0, chord=1, rho=1.225, Uinf=1, Naca="0012", **json.loads(os.environ["FUZZ_CASE_KW"]))
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 50
eng = Engine(0)
eng.set_symmetric(thr if thr else 0)
m = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="full", march=True)
p = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="full", march=False)
t = LUDVM(**kw, verbose=False, engine=eng, precision="f64", history="full", march=False)
scale = np.abs(t.Cl[1:50]).max()
n = 1 + np.arange(m.nt) + np.cumsum(np.r_[0, (t.LEV_shed[1:] != -1)])
for s in range(1, min(m.nt, 45)):
    print(f"step {s:3d}  wake {int(n[s]):4d}  |march - step| {abs(m.Cl[s] - p.Cl[s]) / scale:.2e}   march vs f64 {abs(m.Cl[s] - t.Cl[s]) / scale:.2e}"
          f"   step vs f64 {abs(p.Cl[s] - t.Cl[s]) / scale:.2e}   TEV row diff {np.abs(m.path['TEV'][s] - p.path['TEV'][s]).max():.2e}")
