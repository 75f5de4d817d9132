#!/usr/bin/env python3
"""Bit-level fingerprints of the pair kernels' results, to A/B two builds of the library on one box (GPU box):
    python tools/result_hash.py [--lib path/to/libludvm_hip.so]
prints one JSON line: sha256 of the (u, w) bits of self-interaction calls at sizes that take each symmetric variant
(plain, mixed granularity, quad) and of the direct kernel.
    python tools/result_hash.py --timeloop [--ludvm-module ludvm_amd._ludvm_before]
fingerprints whole runs of the class instead (loads, circulations, LEV shedding, the final wake): the README case marched and
per step in f64 and f32, a Ramesh run, a run with a free-vortex cloud, and config 2's first 6000 steps -- to A/B two versions
of ludvm_amd/ludvm.py (a refactoring must not move a bit)."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--sizes", type=int, nargs="*", default=[20000, 40000, 70000, 200000, 400000, 600001, 1000000])
ap.add_argument("--timeloop", action="store_true")
ap.add_argument("--ludvm-module", default="ludvm_amd.ludvm")
a = ap.parse_args()
from ludvm_amd import _ffi  # noqa: E402
if a.lib:
    _ffi.LIB_PATH = os.path.abspath(a.lib)
import torch  # noqa: E402
from ludvm_amd import Engine  # noqa: E402

eng = Engine(0)
if a.timeloop:
    import importlib
    LUDVM = importlib.import_module(a.ludvm_module).LUDVM

    def fp(sim):
        n = eng.wake_size()
        wx, wz, wg = eng.wake_read(0, n, gamma=True)
        h = hashlib.sha256()
        for v in (sim.Cl, sim.Cd, sim.Cm, sim.LESP, sim.LEV_shed, sim.circulation["TEV"], sim.circulation["LEV"],
                  sim.circulation["bound"], sim.fourier, wx, wz, wg):
            h.update(np.ascontiguousarray(v, dtype=np.float64).tobytes())
        return h.hexdigest()[:16]
    base = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012", verbose=False, engine=eng)
    out = {"module": a.ludvm_module}
    for prec in ("f64", "f32", "f32x2"):
        for march in (True, False):
            for hist in ("full", "sparse"):
                sim = LUDVM(**base, precision=prec, march=march, history=hist, snapshot_steps=[7, 100, 101, 250])
                out[f"readme_{prec}_{'march' if march else 'steps'}_{hist}"] = fp(sim)
    out["ramesh_f64"] = fp(LUDVM(**dict(base, tf=4), method="Ramesh", precision="f64"))
    out["ramesh_f64_steps"] = fp(LUDVM(**dict(base, tf=4), method="Ramesh", precision="f64", march=False))
    rng = np.random.default_rng(9)
    nf = 5000
    xy = np.stack([rng.uniform(1.0, 6.0, nf), rng.uniform(-1.0, 1.0, nf)])
    gam = rng.standard_normal(nf) * 1e-3
    out["free_cloud_f32"] = fp(LUDVM(**dict(base, tf=3), precision="f32", history="sparse", circulation_freevort=gam, xy_freevort=xy,
                                     snapshot_steps=[10, 11]))
    import tempfile
    ck = os.path.join(tempfile.mkdtemp(), "ck.npz")
    sim = LUDVM(**dict(base, tf=6, dt=1e-3), precision="f32", history="sparse", checkpoint_every=2500, checkpoint_path=ck)
    out["cfg2_first_6000_f32"] = fp(sim)
    out["cfg2_resumed_from_5000"] = fp(LUDVM.resume(ck, engine=eng, verbose=False))
    print(json.dumps(out))
    sys.exit(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
out = {"lib": _ffi.LIB_PATH}
for n in a.sizes:
    rng = np.random.default_rng(n)
    x, z = rng.uniform(-10, 0, n).astype(np.float32), rng.uniform(-2, 2, n).astype(np.float32)
    g = (rng.standard_normal(n) / n).astype(np.float32)
    dx, dz, dg = (torch.from_numpy(v).to(dev) for v in (x, z, g))
    du, dw = torch.empty_like(dx), torch.empty_like(dx)
    for sym in (1, 0):
        if sym == 0 and n > 200000:
            continue
        eng.set_symmetric(sym)
        eng.induce_dev(dx.data_ptr(), dz.data_ptr(), dg.data_ptr(), n, dx.data_ptr(), dz.data_ptr(), n, 0.065, du.data_ptr(), dw.data_ptr())
        torch.cuda.synchronize()
        h = hashlib.sha256(du.cpu().numpy().tobytes() + dw.cpu().numpy().tobytes()).hexdigest()[:16]
        out[f"{'sym' if sym else 'direct'}_{n}"] = h
print(json.dumps(out))
