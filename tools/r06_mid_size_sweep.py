#!/usr/bin/env python3
"""ONE sweep, one table, of the roll-up kernel's variants at config 2's mid sizes (GPU box, measurement build).

One roll-up step of the resident wake (ludvm_wake_advect: pair kernel + Euler finisher; an ORDERED sheet spaced like config 2's
wake, fp32 on local origins) per candidate and size, back-to-back steps under sustained load (0.3 s blocks, median of three):
    T4 mixed   pair_sym_f32<4, false, 0>   256-vortex tiles, the bulk by the size rule's waves per item, the end by four
    T4 x4      pair_sym_f32<4, false, 4>   ... every item by four waves
    T8 mixed   pair_sym_f32<8, false, 0>   512-vortex tiles
    T8 x4      pair_sym_f32<8, false, 4>
    direct     pair_f32<.., LOCAL>         every ordered pair
    rule       what the library picks (T = 8 from 36 864 vortices -- 34 816 until this sweep; waves per item by size)  -- must equal the marked candidate
The rule's pick is recomputed here from the constants of pair_sym_kernels.hpp (sym_geometry_t) and marked with *.
    python tools/r06_mid_size_sweep.py [sizes in thousands ...]  > profiles/r06_mid_size_variant_table.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LUDVM_HIP_LIB", os.path.join(ROOT, "ludvm_amd", "csrc", "libludvm_hip_exp.so"))
from ludvm_amd import Engine  # noqa: E402

K_TARGET_WAVES, K_MAX_SPLIT, K_MAX_RSPLIT, K_MIN_ITEMS, K_T8_MIN_N = 8 * 65536, 64, 4, 10500, int(os.environ.get('T8_MIN_N', '36864'))


def rule_pick(n):
    """(T, 'mixed' | 'x4' | 'x2' | 'x1') as sym_tile_t / sym_geometry_t choose for n vortices (tune_* = 0)."""
    T = 8 if n >= K_T8_MIN_N else 4
    W = 64 * T
    nt = max(1, (n + W - 1) // W)
    dmax = (nt - 1) // 2
    dtot = dmax + (1 if (nt % 2 == 0 and nt > 1) else 0)
    ys = K_MAX_SPLIT if nt <= K_TARGET_WAVES // K_MAX_SPLIT else (K_TARGET_WAVES + nt - 1) // nt
    ys = max(1, min(ys, K_MAX_SPLIT, dtot))
    if dtot > 0:
        per0 = (dtot + ys - 1) // ys
        ys = (dtot + per0 - 1) // per0
    rs = 1
    while rs < K_MAX_RSPLIT and nt * ys * rs < K_MIN_ITEMS:
        rs *= 2
    return T, ("mixed" if rs < K_MAX_RSPLIT else "x4")


CANDS = [("T4 mixed", 4, -1), ("T4 x4", 4, 4), ("T8 mixed", 8, -1), ("T8 x4", 8, 4), ("direct", 0, 0), ("rule", 0, 0)]
if os.environ.get("SWEEP_MORE"):            # two and one waves per item too (not among VERDICT r5 item 4's candidates)
    CANDS = CANDS[:4] + [("T4 x2", 4, 2), ("T8 x2", 8, 2), ("T8 x1", 8, 1)] + CANDS[4:]
sizes = [int(a) * 1000 for a in sys.argv[1:]] or [12000, 16000, 20000, 24000, 28000, 32000, 33000, 34000, 35000, 36000, 37000, 38000, 39000,
                                                   40000, 41000, 42000, 43000, 44000, 45000, 46000, 48000, 52000]
eng = Engine(0)
rng = np.random.default_rng(1)
fx, fz, fg = np.linspace(-30.9, -30.0, 80), np.zeros(80), rng.standard_normal(80) / 100
secs = float(os.environ.get("SWEEP_SECONDS", "0.3"))
table = {}
for n in sizes:
    x = -30.0 + 1e-3 * np.arange(n) + 1e-4 * rng.standard_normal(n)           # one vortex per Uinf dt, stored along the sheet
    z = 0.3 * np.sin(0.7 * x) + 1e-3 * rng.standard_normal(n)
    g = rng.standard_normal(n) * 1e-3
    for name, T, R in CANDS:
        eng.set_symmetric(0 if name == "direct" else 2)
        eng.set_sym_tuning(T, R)
        eng.wake_clear()
        eng.wake_append(x, z, g)
        reps = max(5, int(secs * 8.5e12 / (n * n)))
        for _ in range(3):
            eng.wake_advect(1e-6, fx, fz, fg, 1.3e-3, precision="f32")
        eng.synchronize()
        blocks = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.wake_advect(1e-6, fx, fz, fg, 1.3e-3, precision="f32")
            eng.synchronize()
            blocks.append((time.perf_counter() - t0) / reps * 1e6)
        table[(n, name)] = sorted(blocks)[1]
    print(f"# n={n} done", file=sys.stderr, flush=True)
eng.set_sym_tuning(0, 0)
eng.set_symmetric(1)

print(__doc__.split("    python tools")[0].rstrip())
print()
print("microseconds per roll-up step (median of three 0.3 s blocks); * = the rule's pick, ! = a candidate more than 2 % faster than it")
print(f"{'vortices':>9} | " + " | ".join(f"{name:>10}" for name, _, _ in CANDS) + " | rule = pick?  fastest")
worst = 0.0
for n in sizes:
    T, how = rule_pick(n)
    pick = f"T{T} {how}"
    tp = table[(n, pick)]
    cells = []
    for name, _, _ in CANDS:
        v = table[(n, name)]
        mark = "*" if name == pick else ("!" if (name not in ("rule",) and v < tp / 1.02) else " ")
        cells.append(f"{v:9.1f}{mark}")
    sym_best = min((table[(n, c)], c) for c, _, _ in CANDS if c not in ("direct", "rule"))
    allbest = min((table[(n, c)], c) for c, _, _ in CANDS[:-1])
    worst = max(worst, tp / sym_best[0] - 1.0)
    print(f"{n:>9} | " + " | ".join(cells) + f" | {table[(n, 'rule')] / tp - 1.0:+6.1%}       {allbest[1]} ({tp / allbest[0] - 1.0:+.1%} to the pick)")
print()
print(f"largest gap between the rule's pick and the fastest symmetric candidate: {worst:+.1%}")
