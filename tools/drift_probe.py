"""Diagnostic: growth of the difference between the GPU time_loop and the golden config-1 run."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CONFIG1, load_golden
from ludvm_amd import LUDVM, Engine
g = load_golden("g2_config1.npz")
eng = Engine(0)
for prec in ("f32", "f32x2", "f64"):
    sim = LUDVM(**CONFIG1, verbose=False, engine=eng, precision=prec)
    d = {k: np.abs(getattr(sim, k) - g[k]) for k in ("Cl", "Cd", "Cm")}
    print(prec, "LEV_shed same:", np.array_equal(sim.LEV_shed, g["LEV_shed"]))
    for hi in (25, 50, 75, 100, 150, 200, 400):
        print(f"  steps<{hi:3d}: max|dCl| {d['Cl'][:hi].max():.2e} |dCd| {d['Cd'][:hi].max():.2e} |dCm| {d['Cm'][:hi].max():.2e}")
    for s in (1, 2, 10, 50, 100, 400):
        print(f"  TEV pos diff @ {s:3d}: {np.abs(sim.path['TEV'][s] - g[f'TEV_{s}']).max():.2e}")
    print("  period-mean Cl diff (200:):", abs(sim.Cl[200:].mean() - g["Cl"][200:].mean()))
