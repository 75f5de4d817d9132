"""CPU tier: free-vortex cloud generators (ludvm_amd/freevort.py) against the reference's deterministic
cloud stored in the G5 fixture, plus structural checks of the lattice and random generators."""
import numpy as np

from conftest import load_golden
from ludvm_amd import freevort as F


def test_single_vortex_equals_the_reference_cloud():
    g = load_golden("g5_freevort.npz")          # written from the reference's generate_free_single_vortex()
    xy, gam = F.generate_free_single_vortex()
    assert xy.shape == (61, 2) and gam.shape == (61,)
    np.testing.assert_allclose(xy.T, g["xy_freevort"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(gam, g["gamma_freevort"], rtol=0, atol=1e-16)
    assert abs(gam.sum() - 10.0) < 1e-12


def test_lattice_alternates_sign_and_matches_the_reference():
    xy, gam = F.generate_flowfield_vortices()
    per = 6                                       # 1 + 5 point vortices per cloud
    assert len(gam) % per == 0 and xy.shape == (len(gam), 2)
    signs = np.sign(gam[::per])
    ncy = len(np.arange(-3 + 0.2, 2.5 - 0.2 + 0.4, 0.4))
    grid = signs.reshape(-1, ncy)
    assert np.all(grid[:-1] * grid[1:] < 0) and np.all(grid[:, :-1] * grid[:, 1:] < 0)
    g6 = load_golden("g6_generators.npz")
    np.testing.assert_allclose(xy, g6["lattice_xy"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(gam, g6["lattice_gamma"], rtol=0, atol=1e-16)
    xy1, g1 = F.generate_free_single_vortex()
    np.testing.assert_allclose(xy1, g6["single_xy"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(g1, g6["single_gamma"], rtol=0, atol=1e-16)


def test_turbulence_is_seeded_and_separated():
    xy1, g1 = F.generate_flowfield_turbulence(seed=3, vortex_density=0.3)
    xy2, g2 = F.generate_flowfield_turbulence(seed=3, vortex_density=0.3)
    assert np.array_equal(xy1, xy2) and np.array_equal(g1, g2)
    centres = xy1[::6]                            # first point of each cloud sits on its centre
    d = np.hypot(centres[:, None, 0] - centres[None, :, 0], centres[:, None, 1] - centres[None, :, 1])
    d[np.diag_indices_from(d)] = np.inf
    assert d.min() >= 0.4 - 1e-12
    assert set(np.unique(np.sign(g1))) <= {-1.0, 1.0}
