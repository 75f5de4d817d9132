#!/usr/bin/env python3
"""One configuration of tools/fuzz_march.py looked at closely: distance of every fp32 path from the float64 run per window of
steps (march overlapped / march serial / per-step path, fp32 on local origins and hi+lo).  Run on the GPU box.
    python tools/fuzz_case_probe.py   (the seed-9 / case-0 configuration of profiles/r02_fuzz_final_code.txt)
    FUZZ_CASE_KW="$(cat tests/tools/cases/fuzz_march_seed4_case4.json)" python tools/fuzz_case_probe.py   (another configuration:
    the one analysed in profiles/r03_fuzz_final_code.txt)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LUDVM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ludvm_amd", "csrc", "libludvm_hip_exp.so"))  # measurement build: forced variants / A-B switches
KW = dict(t0=0, tf=2.58, dt=0.02, chord=1, rho=1.225, Uinf=1, Npoints=121, Ncoeffs=16, LESPcrit=0.2611018525180467, Naca="0012",
          alpha_m=2.775340829201788, alpha_max=22.050089628899826, k=2.3053742771281676, phi=154.87085685291925,
          h_max=1.3773564435731436, method="Faure")


if os.environ.get("FUZZ_CASE_KW"):        # another configuration: the "kw" object of a fuzz_march.py failure line (JSON)
    KW = dict(t0=0, chord=1, rho=1.225, Uinf=1, Naca="0012", **json.loads(os.environ["FUZZ_CASE_KW"]))


def run(precision, march, threshold):
    from ludvm_amd import LUDVM, Engine
    eng = Engine(0)
    eng.set_symmetric(threshold)
    sim = LUDVM(**KW, verbose=False, engine=eng, precision=precision, history="full", march=march)
    out = (sim.Cl.copy(), (sim.LEV_shed != -1).copy())
    eng.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:          # child: one run, printed as JSON (LUDVM_MARCH_OVERLAP is read when the library loads)
        prec, march, thr = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3])
        cl, shed = run(prec, march, thr)
        print(json.dumps({"cl": cl.tolist(), "shed": shed.tolist()}))
        sys.exit(0)

    def child(prec, march, thr, overlap=True):
        env = dict(os.environ)
        env["LUDVM_MARCH_OVERLAP"] = "1" if overlap else "0"
        r = subprocess.run([sys.executable, __file__, prec, "1" if march else "0", str(thr)], env=env, capture_output=True, text=True, check=True)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return np.array(d["cl"]), np.array(d["shed"])

    ref, ref_shed = child("f64", False, 1)
    scale = np.abs(ref[1:50]).max()
    rows = []
    for name, args in (("f32 per-step, symmetric from 50", ("f32", False, 50)), ("f32 march overlapped, from 50", ("f32", True, 50)),
                       ("f32 march serial, from 50", ("f32", True, 50, False)), ("f32 per-step, direct kernel", ("f32", False, 0)),
                       ("f32 march, direct kernel", ("f32", True, 0)), ("f32x2 per-step, from 50", ("f32x2", False, 50)),
                       ("f32x2 march overlapped, from 50", ("f32x2", True, 50))):
        cl, shed = child(*args)
        d = np.abs(cl - ref) / scale
        first = int(np.argmax(shed != ref_shed)) if (shed != ref_shed).any() else -1
        rows.append({"run": name, "vs_f64": {str(k): float(d[1:k].max()) for k in (15, 25, 35, 50, 75, 100)}, "shedding_departs_at": first})
        print(json.dumps(rows[-1]), flush=True)
