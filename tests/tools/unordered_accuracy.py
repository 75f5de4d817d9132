#!/usr/bin/env python3
"""fp32 accuracy on UNORDERED input (GPU box): N vortices uniformly random in a 10 x 4 box centred at x = -55 -- the order a
caller's array or a generate_flowfield_turbulence cloud (LUDVM.py:98-130) has, not a shed wake's -- v_core = 1.3e-3, through
the host-pointer entry (ludvm_induce_f64, what LUDVM.induced_velocity calls) and through the resident wake (wake_append +
ludvm_wake_advect), symmetric and direct kernel, sampled targets against the float64 C oracle.  Prints max error / max|u|.
    python tests/tools/unordered_accuracy.py [N ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ludvm_amd import Engine  # noqa: E402
from oracle import c_oracle  # noqa: E402


def cloud(n, seed=77):
    rng = np.random.default_rng(seed)
    return rng.uniform(-60.0, -50.0, n), rng.uniform(-2.0, 2.0, n), rng.standard_normal(n) * 1e-2


def main():
    eng = Engine(0)
    vc = 1.3e-3
    for n in [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]:
        x, z, g = cloud(n)
        sel = np.random.default_rng(3).choice(n, 512, replace=False)
        ur, wr = c_oracle.induced_velocity(g, x, z, x[sel], z[sel], vc)
        scale = max(np.abs(ur).max(), np.abs(wr).max())
        # max|u| over ALL targets from the hi+lo run (good to 1e-6; the oracle cannot do N^2): the tier's normalisation.
        # `scale` (the 512 samples' own maximum) is smaller, i.e. stricter; both are printed
        eng.set_symmetric(1)
        ua, wa = eng.induce(g, x, z, x, z, vc, precision="f32x2")
        scale_all = max(np.abs(ua).max(), np.abs(wa).max())
        rec = {"n": n, "max_abs_u_sampled": scale, "max_abs_u_all": scale_all}
        for sym in (1, 0):
            eng.set_symmetric(sym)
            kname = "symmetric" if sym else "direct"
            for prec in ("f32", "f32x2"):
                t0 = time.perf_counter()
                u, w = eng.induce(g, x, z, x, z, vc, precision=prec)
                el = time.perf_counter() - t0
                rec[f"induce_{kname}_{prec}"] = float(max(np.abs(u[sel] - ur).max(), np.abs(w[sel] - wr).max()) / scale)
                rec[f"induce_{kname}_{prec}_s"] = round(el, 4)
            # disjoint targets (the direct kernel with sources != targets): the sampled points as a separate target array
            u, w = eng.induce(g, x, z, x[sel].copy(), z[sel].copy(), vc, precision="f32")
            rec[f"induce_{kname}_targets_f32"] = float(max(np.abs(u - ur).max(), np.abs(w - wr).max()) / scale)
            eng.wake_clear()
            eng.wake_append(x, z, g)
            u, w = eng.wake_advect(2.0 ** -10, [], [], [], vc, precision="f32", return_velocity=True)
            rec[f"wake_advect_{kname}_f32_callers_order"] = float(max(np.abs(u[sel] - ur).max(), np.abs(w[sel] - wr).max()) / scale)
            if hasattr(eng, "spatial_order"):
                # what LUDVM.time_loop does with a cloud of free vortices: store it in the order the engine names, and take
                # hi+lo positions when even that order leaves the classes too wide for the core
                order, reordered, extent = eng.spatial_order(x, z, with_extent=True)
                slot = np.empty(n, np.int64)
                slot[order] = np.arange(n)
                prec = "f32x2" if extent > (150 if reordered else 300) * vc else "f32"
                eng.wake_clear()
                eng.wake_append(x[order], z[order], g[order])
                u, w = eng.wake_advect(2.0 ** -10, [], [], [], vc, precision=prec, return_velocity=True)
                rec[f"wake_advect_{kname}_engine_order_{prec}"] = float(max(np.abs(u[slot[sel]] - ur).max(),
                                                                        np.abs(w[slot[sel]] - wr).max()) / scale)
                rec["mean_class_extent_over_vcore"] = extent / vc
        eng.set_symmetric(1)
        rec["normalised_by_max_over_all_targets"] = {k: v * scale / scale_all for k, v in rec.items()
                                                     if k.startswith(("induce", "wake")) and not k.endswith("_s")}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
