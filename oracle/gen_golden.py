#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing and running the reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where /root/reference exists; the GPU box
never sees the reference, only the .npz files this script writes (inputs + expected outputs).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Fixture groups:
  G1 kernel KATs      -- LUDVM.induced_velocity (LUDVM.py:549-570) called unbound on seeded inputs;
                         needs no stand-in of any kind.
  G2 config-1 run     -- the README example (LUDVM.py:161-162): loads, Fourier coefficients, LESP,
                         circulations, wake snapshots.
  G3 boundary trace   -- every induced_velocity call (arguments and returns) of selected steps.
  G4 flowfield        -- coarse grid at three time steps (LUDVM.py:1186-1298).
  G5 variants         -- method='Ramesh', user free vortices, alpha_m != 0.
  G6 generators       -- the deterministic free-vortex cloud builders (LUDVM.py:53-96).
  G7 config-2 regime  -- BASELINE config 2's parameters (dt = 1e-3, v_core = 1.3e-3) run by the reference itself for the
                         first 1500 steps (the full 50 000 do not fit its dense history): loads, circulations, LESP, the
                         onset of LEV shedding (step 1335) and three wake rows.
G2-G5 run the unmodified reference class with oracle/airfoils_standin on sys.path (zero camber,
valid for the symmetric NACA0012 all BASELINE configs use).
"""
import os
import sys
import io
import types
import contextlib
import warnings

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "airfoils_standin"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

warnings.simplefilter("ignore", DeprecationWarning)
import LUDVM as REF  # noqa: E402  (the reference module)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ------------------------------------------------------------------------------------------- G1
def g1_kernel_kats():
    rng = np.random.default_rng(20260101)
    cases = {}

    def wake_like(n, offset=0.0):
        return rng.uniform(-10, 0, n) + offset, rng.uniform(-2, 2, n), rng.standard_normal(n)

    def add(name, g, xw, zw, xp, zp, v_core, viscous=True):
        ns = types.SimpleNamespace(v_core=v_core)
        u, w = REF.LUDVM.induced_velocity(ns, g, xw, zw, xp, zp, viscous=viscous)
        cases[name] = dict(g=np.asarray(g), xw=np.asarray(xw, float), zw=np.asarray(zw, float),
                           xp=np.asarray(xp, float), zp=np.asarray(zp, float),
                           v_core=float(v_core), viscous=bool(viscous), u=u, w=w)

    for vc, tag in ((0.065, "vc065"), (1.3e-3, "vc0013")):
        # single pair
        add(f"p1x1_{tag}", np.array([0.7]), np.array([-1.0]), np.array([0.25]), np.array([-0.4]), np.array([-0.1]), vc)
        # unit new TEV -> 80 chord points, integer circulation [1] (LUDVM.py:751)
        xa = np.linspace(-1.0, 0.0, 80) - 3.0
        za = 0.05 * np.sin(np.linspace(0, 3, 80))
        add(f"p80x1_int_{tag}", np.array([1]), np.array([0.012 - 3.0]), np.array([0.001]), xa, za, vc)
        # wake -> chord
        x, z, g = wake_like(603)
        add(f"p80x603_{tag}", g, x, z, xa, za, vc)
        # TEV slice of the wake as targets (targets are a subset of the sources: self pairs)
        x, z, g = wake_like(604)
        add(f"p400x604_self_{tag}", g, x, z, x[:400], z[:400], vc)
        add(f"p1x604_{tag}", g, x, z, x[603:], z[603:], vc)
        # odd sizes, disjoint sets
        xs, zs, gs = wake_like(1023)
        xt, zt, _ = wake_like(257)
        add(f"p257x1023_{tag}", gs, xs, zs, xt, zt, vc)
        # coordinates offset by -50 (late-time wake)
        xs, zs, gs = wake_like(1023, offset=-50.0)
        add(f"p257x1023_off50_{tag}", gs, xs, zs, xs[:257] + 1e-3, zs[:257] - 2e-3, vc)
        # non-contiguous views, as time_loop passes them (path[...][i,0,:k])
        big = rng.uniform(-5, 0, (3, 2, 700))
        gam = rng.standard_normal(700)
        add(f"p300x650_strided_{tag}", gam[:650], big[1, 0, :650], big[1, 1, :650], big[2, 0, :300], big[2, 1, :300], vc)
    # inviscid on disjoint sets (viscous != True -> v_core = 0, LUDVM.py:562-563)
    xs, zs, gs = wake_like(333)
    xt, zt, _ = wake_like(129)
    add("p129x333_inviscid", gs, xs, zs, xt, zt, 0.065, viscous=False)
    # inviscid with a coincident pair: the reference returns NaN there (0/0)
    with np.errstate(all="ignore"):
        add("p3x3_inviscid_self", np.array([1.0, -2.0, 0.5]), np.array([0.0, 1.0, 2.0]), np.array([0.0, 0.5, -0.5]),
            np.array([0.0, 1.5, 2.0]), np.array([0.0, 0.25, -0.5]), 0.065, viscous=False)
    # a medium all-pairs case
    x, z, g = wake_like(2048)
    add("p2048x2048_self_vc065", g / 2048, x, z, x, z, 0.065)

    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "g1_kernel_kats.npz"), **flat)
    print("G1:", len(cases), "cases")


# ------------------------------------------------------------------------------------------- runs
CONFIG1 = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")


class Spy:
    """Records every induced_velocity call of selected time steps (G3)."""

    def __init__(self, steps):
        self.steps = set(steps)
        self.calls = []
        self.step = None


def run_reference(kwargs, spy=None):
    cls = REF.LUDVM
    if spy is None:
        return quiet(cls, **kwargs)
    orig_iv = cls.induced_velocity
    orig_dw = cls.airfoil_downwash

    def iv(self, circulation, xw, zw, xp, zp, viscous=True):
        u, w = orig_iv(self, circulation, xw, zw, xp, zp, viscous)
        # the current step is recovered from the wake-copy already done for row i (LUDVM.py:664-666):
        # spy.step is set by the downwash wrapper below, which receives i explicitly
        if spy.step in spy.steps:
            spy.calls.append(dict(step=spy.step, g=np.array(circulation), xw=np.array(xw), zw=np.array(zw),
                                  xp=np.array(xp), zp=np.array(zp), viscous=bool(viscous), u=u.copy(), w=w.copy()))
        return u, w

    def dw(self, circulation, xw, zw, i):
        spy.step = i
        return orig_dw(self, circulation, xw, zw, i)

    cls.induced_velocity = iv
    cls.airfoil_downwash = dw
    try:
        return quiet(cls, **kwargs)
    finally:
        cls.induced_velocity = orig_iv
        cls.airfoil_downwash = orig_dw


def pack_run(sim, snaps):
    d = dict(
        nt=sim.nt, itev=sim.itev, ilev=sim.ilev, v_core=sim.v_core,
        Cl=sim.Cl, Cd=sim.Cd, Cm=sim.Cm, Cn=sim.Cn, Cs=sim.Cs, Ct=sim.Ct,
        Fn=sim.Fn, Fs=sim.Fs, L=sim.L, D=sim.D, T=sim.T, M=sim.M,
        LESP=sim.LESP, LESP_prev=sim.LESP_prev, LEV_shed=sim.LEV_shed,
        circ_TEV=sim.circulation["TEV"], circ_LEV=sim.circulation["LEV"], circ_bound=sim.circulation["bound"],
        circ_FREE=np.asarray(sim.circulation["FREE"], float), circ_IC=float(sim.circulation["IC"]),
        circ_airfoil_last=sim.circulation["airfoil"][sim.itev], fourier=sim.fourier,
        alpha=sim.alpha, alpha_dot=sim.alpha_dot, h_dot=sim.h_dot, t=sim.t,
        airfoil_x=sim.airfoil["x"], airfoil_theta_panel=sim.airfoil["theta_panel"],
        path_airfoil_last=sim.path["airfoil"][-1], path_gamma_points_1=sim.path["airfoil_gamma_points"][1],
        snap_steps=np.array(snaps),
    )
    for s in snaps:
        d[f"TEV_{s}"] = sim.path["TEV"][s]
        d[f"LEV_{s}"] = sim.path["LEV"][s]
        d[f"FREE_{s}"] = sim.path["FREE"][s]
    return d


def g2_g3_g4_config1():
    spy = Spy(steps=[1, 2, 3, 4, 5, 100, 400])
    sim = run_reference(CONFIG1, spy)
    d = pack_run(sim, [1, 2, 10, 50, 100, 400])
    np.savez_compressed(os.path.join(OUT, "g2_config1.npz"), **d)
    print("G2: nt", sim.nt, "itev", sim.itev, "ilev", sim.ilev, "Cl[-3:]", sim.Cl[-3:])

    flat = {"ncalls": len(spy.calls)}
    for k, c in enumerate(spy.calls):
        for key, v in c.items():
            flat[f"{k}/{key}"] = v
    np.savez_compressed(os.path.join(OUT, "g3_boundary_trace.npz"), **flat)
    print("G3:", len(spy.calls), "calls")

    tsteps = [0, 50, 200]
    quiet(sim.flowfield, xmin=-10, xmax=0, zmin=-4, zmax=4, dr=0.25, tsteps=tsteps)
    np.savez_compressed(os.path.join(OUT, "g4_flowfield.npz"), tsteps=np.array(tsteps), dr=0.25,
                        box=np.array([-10.0, 0.0, -4.0, 4.0]), x_ff=sim.x_ff, z_ff=sim.z_ff,
                        u_ff=sim.u_ff, w_ff=sim.w_ff, ome_ff=sim.ome_ff)
    print("G4: grid", sim.x_ff.shape)


def g5_variants():
    # (a) Ramesh (Newton) method, short run
    kw = dict(CONFIG1, tf=2, method="Ramesh")
    sim = run_reference(kw)
    np.savez_compressed(os.path.join(OUT, "g5_ramesh.npz"), **pack_run(sim, [1, 10, sim.nt - 1]))
    print("G5a Ramesh: nt", sim.nt, "ilev", sim.ilev)
    # (b) user free vortices: the reference's own deterministic cloud (LUDVM.py:53-71)
    xy, gam = REF.generate_free_single_vortex()
    kw = dict(CONFIG1, tf=5, circulation_freevort=gam, xy_freevort=np.transpose(xy))
    sim = run_reference(kw)
    d = pack_run(sim, [1, 10, sim.nt - 1])
    d["xy_freevort"] = np.transpose(xy)
    d["gamma_freevort"] = gam
    np.savez_compressed(os.path.join(OUT, "g5_freevort.npz"), **d)
    print("G5b free vortices: n_free", sim.n_freevort, "ilev", sim.ilev)
    # (c) non-zero mean pitch
    kw = dict(CONFIG1, tf=5, alpha_m=5, alpha_max=15)
    sim = run_reference(kw)
    np.savez_compressed(os.path.join(OUT, "g5_alpham.npz"), **pack_run(sim, [1, 10, sim.nt - 1]))
    print("G5c alpha_m: ilev", sim.ilev)


def g6_generators():
    """Deterministic free-vortex generators (LUDVM.py:53-96)."""
    xy1, g1 = REF.generate_free_single_vortex()
    xy2, g2 = REF.generate_flowfield_vortices()
    np.savez_compressed(os.path.join(OUT, "g6_generators.npz"), single_xy=xy1, single_gamma=g1, lattice_xy=xy2,
                        lattice_gamma=g2)
    print("G6: single", xy1.shape, "lattice", xy2.shape)


def g7_config2_regime():
    """The reference at config 2's parameters over the first 1500 steps (LUDVM.py:597-1171 at dt = 1e-3)."""
    import time
    kw = dict(CONFIG1, dt=1e-3, tf=1.5)
    t0 = time.time()
    sim = run_reference(kw)
    shed = sim.LEV_shed != -1
    d = dict(nt=sim.nt, itev=sim.itev, ilev=sim.ilev, v_core=sim.v_core, Cl=sim.Cl, Cd=sim.Cd, Cm=sim.Cm, LESP=sim.LESP,
             LEV_shed=sim.LEV_shed, circ_TEV=sim.circulation["TEV"], circ_LEV=sim.circulation["LEV"],
             circ_bound=sim.circulation["bound"], circ_IC=float(sim.circulation["IC"]), fourier4=sim.fourier[:, :, :4],
             first_lev_step=int(np.argmax(shed)) if shed.any() else -1)
    for s_ in (300, 1000, 1500):
        d[f"TEV_{s_}"] = sim.path["TEV"][s_][:, :s_ + 1]
        d[f"LEV_{s_}"] = sim.path["LEV"][s_][:, :max(1, sim.ilev + 1)]
    np.savez_compressed(os.path.join(OUT, "g7_config2_first1500.npz"), **d)
    print("G7: nt", sim.nt, "itev", sim.itev, "ilev", sim.ilev, "first LEV at step", d["first_lev_step"],
          "reference wall %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        g7_config2_regime()
        sys.exit(0)
    g1_kernel_kats()
    g2_g3_g4_config1()
    g5_variants()
    g6_generators()
    g7_config2_regime()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total fixture bytes:", tot)
