"""CPU tier: host logic of the drop-in LUDVM class (ludvm_amd/ludvm.py) against the golden vectors,
with the pair sums served by tests/fake_engine.FakeEngine (oracle arithmetic, float64).  The GPU
tier (tests/test_gpu_*.py) repeats these comparisons through the real HIP engine."""
import numpy as np
import pytest

from conftest import CONFIG1, load_golden
from fake_engine import FakeEngine
from ludvm_amd.ludvm import LUDVM, SparseHistory


@pytest.fixture(scope="module")
def sim1():
    return LUDVM(**CONFIG1, verbose=False, engine=FakeEngine())


def _compare(sim, g, tol_loads, tol_early_pos, early_steps):
    assert sim.nt == int(g["nt"]) and sim.itev == int(g["itev"]) and sim.ilev == int(g["ilev"])
    np.testing.assert_array_equal(sim.LEV_shed, g["LEV_shed"])
    for name in ("Cl", "Cd", "Cm", "Fn", "Fs", "M", "LESP", "LESP_prev"):
        np.testing.assert_allclose(getattr(sim, name), g[name], rtol=0, atol=tol_loads, err_msg=name)
    np.testing.assert_allclose(sim.circulation["TEV"], g["circ_TEV"], rtol=0, atol=tol_loads)
    np.testing.assert_allclose(sim.circulation["LEV"], g["circ_LEV"], rtol=0, atol=tol_loads)
    np.testing.assert_allclose(sim.circulation["bound"], g["circ_bound"], rtol=0, atol=tol_loads)
    np.testing.assert_allclose(sim.fourier, g["fourier"], rtol=0, atol=10 * tol_loads)
    for s in g["snap_steps"]:
        if s <= early_steps:
            for key in ("TEV", "LEV", "FREE"):
                np.testing.assert_allclose(sim.path[key][s], g[f"{key}_{s}"], rtol=0, atol=tol_early_pos,
                                           err_msg=f"{key}@{s}")


def test_config1_matches_reference(sim1, g2):
    # The host side uses matrix forms (trapezoid weights, cumsum) instead of the reference's loops, so
    # results differ from it by rounding only; the run is chaotic, which turns 1e-16 into
    # ~1e-6 on the loads by step 400 and O(0.1) on late wake positions.  Tier T3 tolerance: 1e-5.
    _compare(sim1, g2, tol_loads=1e-5, tol_early_pos=1e-9, early_steps=100)
    # rows of the dense history include the phantom LEV slot of non-shedding steps
    np.testing.assert_allclose(sim1.path["LEV"][2], g2["LEV_2"], rtol=0, atol=1e-12)


def test_config1_uses_one_point_sum_per_step(sim1):
    # dense history records every row, so every step takes the two-call path: one fused chord call (wake sum +
    # unit TEV / candidate LEV) and one fused roll-up
    eng = sim1.engine
    assert eng.calls["chord"] == 400 and eng.calls["advect"] == 400
    assert eng.calls["induce"] == 0 and eng.calls["points"] == 0


def test_sparse_history_takes_the_one_round_trip_path(sim1):
    sp = LUDVM(**CONFIG1, verbose=False, engine=FakeEngine(), history="sparse", snapshot_steps=[100, 250])
    eng = sp.engine
    # 400 steps, rows recorded at 100, 250 and 400: those three use the two-call path, and so does the step
    # after each recorded one (its sums were not produced by a wake_step) -- step 1 likewise
    assert eng.calls["step"] == 400 - 3 and eng.calls["advect"] == 400
    # chord sums: 397 inside wake_step (counted there) + the separate calls of steps 1, 101, 251
    assert eng.calls["chord"] == 397 + 3
    # same arithmetic; the dense path additionally carries the zero-strength phantom LEV slot, which regroups
    # the oracle's pairwise sums: ulp-level differences, amplified by the chaotic wake to ~1e-6 by step 400
    np.testing.assert_allclose(sp.Cl[:100], sim1.Cl[:100], rtol=0, atol=1e-10)
    np.testing.assert_allclose(sp.Cl, sim1.Cl, rtol=0, atol=1e-5)
    np.testing.assert_allclose(sp.path["TEV"][100], sim1.path["TEV"][100, :, :100], rtol=0, atol=1e-9)
    assert np.array_equal(sp.LEV_shed, sim1.LEV_shed)


def test_public_methods_match_reference_signatures(sim1, g1_cases):
    c = g1_cases["p80x603_vc065"]
    sim1_v = sim1.v_core
    assert sim1_v == 0.065
    u, w = sim1.induced_velocity(c["g"], c["xw"], c["zw"], c["xp"], c["zp"])
    np.testing.assert_array_equal(u, c["u"])
    assert u.dtype == np.float64 and u.shape == (80,)
    # airfoil_downwash(circulation, xw, zw, i) -> W[Npoints-1]
    W = sim1.airfoil_downwash(c["g"], c["xw"], c["zw"], 5)
    assert W.shape == (80,)


def test_flowfield_matches_reference(sim1):
    g4 = load_golden("g4_flowfield.npz")
    xmin, xmax, zmin, zmax = g4["box"]
    sim1.flowfield(xmin=xmin, xmax=xmax, zmin=zmin, zmax=zmax, dr=float(g4["dr"]), tsteps=list(g4["tsteps"]))
    np.testing.assert_array_equal(sim1.x_ff, g4["x_ff"])
    np.testing.assert_array_equal(sim1.z_ff, g4["z_ff"])
    # step 0 and 50 are before the chaotic divergence bites; step 200 is compared loosely
    for k, tol in ((0, 1e-10), (1, 1e-6)):
        np.testing.assert_allclose(sim1.u_ff[k], g4["u_ff"][k], rtol=0, atol=tol)
        np.testing.assert_allclose(sim1.w_ff[k], g4["w_ff"][k], rtol=0, atol=tol)
        np.testing.assert_allclose(sim1.ome_ff[k], g4["ome_ff"][k], rtol=0, atol=100 * tol)
    assert sim1.u_ff.shape == g4["u_ff"].shape


@pytest.mark.parametrize("fixture,kwargs", [
    ("g5_ramesh.npz", dict(tf=2, method="Ramesh")),
    ("g5_alpham.npz", dict(tf=5, alpha_m=5, alpha_max=15)),
])
def test_variants(fixture, kwargs):
    g = load_golden(fixture)
    sim = LUDVM(**dict(CONFIG1, **kwargs), verbose=False, engine=FakeEngine())
    _compare(sim, g, tol_loads=1e-7, tol_early_pos=1e-9, early_steps=10)


def test_free_vortices():
    g = load_golden("g5_freevort.npz")
    sim = LUDVM(**dict(CONFIG1, tf=5, circulation_freevort=g["gamma_freevort"], xy_freevort=g["xy_freevort"]),
                verbose=False, engine=FakeEngine())
    assert sim.n_freevort == 61
    _compare(sim, g, tol_loads=1e-6, tol_early_pos=1e-9, early_steps=10)


def test_sparse_history_rows_equal_dense_rows(sim1):
    sp = LUDVM(**CONFIG1, verbose=False, engine=FakeEngine(), history="sparse", snapshot_steps=[1, 10, 49, 50])
    assert isinstance(sp.path["TEV"], SparseHistory)
    assert sp.path["TEV"].steps() == [0, 1, 10, 49, 50, 400]
    for s in (1, 10, 50):
        np.testing.assert_allclose(sp.path["TEV"][s], sim1.path["TEV"][s, :, :s], rtol=0, atol=1e-9)
        k = min(3, s)
        np.testing.assert_allclose(sp.path["TEV"][s, 0, :k], sim1.path["TEV"][s, 0, :k], rtol=0, atol=1e-9)
    with pytest.raises(KeyError):
        sp.path["TEV"][7]
    np.testing.assert_allclose(sp.Cl[:100], sim1.Cl[:100], rtol=0, atol=1e-9)
    # flow field from sparse rows needs rows s-1 and s
    sp.verbose = False
    sp.flowfield(dr=0.5, tsteps=[0, 50])
    sim1.flowfield(dr=0.5, tsteps=[0, 50])
    np.testing.assert_allclose(sp.u_ff, sim1.u_ff, rtol=0, atol=1e-8)
    with pytest.raises(KeyError):
        sp.flowfield(dr=0.5, tsteps=[30])


def test_constructor_validation():
    with pytest.raises(ValueError):
        LUDVM(**CONFIG1, engine=FakeEngine(), precision="bf16")
    with pytest.raises(ValueError):
        LUDVM(**CONFIG1, engine=FakeEngine(), history="ring")
    s = LUDVM(**CONFIG1, engine=FakeEngine(), verbose=False, run=False)
    assert s.nt == 401 and s.history == "full" and s.path["airfoil"].shape == (401, 2, 81)
    with pytest.raises(ValueError):
        LUDVM(**dict(CONFIG1, method="Newton", tf=0.2), engine=FakeEngine(), verbose=False)


@pytest.mark.parametrize("history", ["full", "sparse"])
def test_checkpoint_resume_is_bitwise_with_a_deterministic_engine(tmp_path, sim1, history):
    ck = str(tmp_path / "run.npz")
    kw = dict(CONFIG1, tf=10)
    a = LUDVM(**kw, verbose=False, engine=FakeEngine(), history=history, snapshot_steps=[50, 120])
    b = LUDVM(**kw, verbose=False, engine=FakeEngine(), history=history, snapshot_steps=[50, 120],
              checkpoint_every=70, checkpoint_path=ck)           # writes at steps 70 and 140
    c = LUDVM.resume(ck, engine=FakeEngine(), verbose=False)     # continues from step 141
    for name in ("Cl", "Cd", "Cm", "LESP", "LESP_prev", "fourier", "LEV_shed"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert np.array_equal(getattr(a, name), getattr(c, name)), name
    for key in ("TEV", "LEV", "bound", "airfoil"):
        assert np.array_equal(a.circulation[key], c.circulation[key]), key
    assert (a.itev, a.ilev) == (c.itev, c.ilev)
    for s in (50, 120, 200):
        for key in ("TEV", "LEV", "FREE"):
            ra, rc = a.path[key][s], c.path[key][s]
            assert np.array_equal(np.asarray(ra), np.asarray(rc)), (key, s)
    with pytest.raises(ValueError):
        LUDVM(**kw, verbose=False, engine=FakeEngine(), checkpoint_every=10)


def test_dat_section_and_sin_motion(tmp_path):
    # a symmetric section in Selig format (upper surface TE -> LE, then lower LE -> TE): mean line ~ 0
    xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, 41)))
    yt = 0.6 * (0.2969 * np.sqrt(xs) - 0.126 * xs - 0.3516 * xs**2 + 0.2843 * xs**3 - 0.1015 * xs**4)
    pts = np.r_[np.c_[xs[::-1], yt[::-1]], np.c_[xs[1:], -yt[1:]]]
    dat = tmp_path / "sym.dat"
    dat.write_text("SYMMETRIC TEST SECTION\n" + "\n".join(f"{a:.6f} {b:.6f}" for a, b in pts) + "\n")
    with pytest.warns(RuntimeWarning, match="UNPINNED"):          # a .dat section: said at run time, once per construction
        s = LUDVM(**dict(CONFIG1, tf=0.5, Naca=None, foil_filename=str(dat)), engine=FakeEngine(), verbose=False)
    assert np.abs(s.airfoil["eta"]).max() < 1e-6 and s.airfoil["x"].shape == (81,)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # the pinned symmetric sections do not warn
        ref = LUDVM(**dict(CONFIG1, tf=0.5), engine=FakeEngine(), verbose=False)
    np.testing.assert_allclose(s.Cl, ref.Cl, rtol=0, atol=1e-4)
    # cambered NACA digits run (parity unpinned) and give a lifting mean line -- with the warning
    with pytest.warns(RuntimeWarning, match="LUDVM.py:301-335") as rec:
        c = LUDVM(**dict(CONFIG1, tf=0.5, Naca="2412"), engine=FakeEngine(), verbose=False)
    assert len([w for w in rec if "UNPINNED" in str(w.message)]) == 1
    assert abs(c.airfoil["eta"].max() - 0.02) < 2e-4 and np.all(np.isfinite(c.Cl))
    # motion='sin' (LUDVM.py:417-421)
    m = LUDVM(**dict(CONFIG1, tf=0.5), engine=FakeEngine(), verbose=False, run=False)
    m.motion_sinusoidal(alpha_m=0, alpha_max=10, h_max=1, k=0.2 * np.pi, phi=90, h0=0, x0=0, motion="sin")
    assert abs(m.hpiv[0]) < 1e-15 and abs(m.alpha[0] - np.deg2rad(10)) < 1e-12
    with pytest.raises(ValueError):
        m.motion_sinusoidal(motion="saw")
