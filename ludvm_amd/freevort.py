"""Free-vortex cloud generators (reference LUDVM.py:18-130): inputs for the `circulation_freevort` /
`xy_freevort` constructor arguments.  Each returns `(xyvorts [N, 2], gammavorts [N])`; pass
`xy_freevort=xyvorts.T` as the reference's example does (LUDVM.py:1400).

Differences from the reference, by intent: the random generator takes a `seed`/`rng` so clouds are
reproducible (the reference draws from the global NumPy state, :106-126); nothing is printed.
"""
import numpy as np

__all__ = ["generate_free_vortices", "generate_free_single_vortex", "generate_flowfield_vortices",
           "generate_flowfield_turbulence"]


def generate_free_vortices(nvorts, cvorts, vortradius, layerspervort, npervortlayer, gammapervort):
    """`nvorts` clouds of point vortices centred at `cvorts[n]`: `layerspervort` concentric rings of
    radius 0..`vortradius` holding `npervortlayer[k]` vortices each (LUDVM.py:18-51).

    As in the reference, the circulation of cloud n is spread as gammapervort[n] / (number of point
    vortices generated SO FAR, this cloud included) -- so with several clouds the later ones get weaker
    point vortices (:46); kept for drop-in compatibility."""
    radii = vortradius * np.linspace(0, 1, layerspervort)
    ring = np.concatenate([np.stack([r * np.cos(t), r * np.sin(t)], axis=1)
                           for r, k in zip(radii, npervortlayer)
                           for t in [np.linspace(0, 2 * np.pi, k, endpoint=False)]])
    xy, gam = [], []
    count = 0
    for n in range(nvorts):
        pts = ring + np.asarray(cvorts)[n, :]
        count += len(pts)
        xy.append(pts)
        gam.append(np.full(len(pts), gammapervort[n] / count))
    return np.concatenate(xy), np.concatenate(gam)


def generate_free_single_vortex():
    """One cloud of 61 point vortices, total circulation 10, radius 0.5, centred at (-2.5, -0.5)
    (LUDVM.py:53-71)."""
    return generate_free_vortices(1, np.array([[-2.5, -0.5]]), 0.5, 5, np.array([1, 5, 10, 15, 30]), 10 * np.array([1]))


def generate_flowfield_vortices(vortex_radius=0.2, gamma=0.5, xmin=-5, xmax=0, ymin=-3, ymax=2.5, layerspervort=2,
                                npervortlayer=np.array([1, 5]), centers_separation_factor=1):
    """Taylor-Green-like lattice of counter-rotating clouds (LUDVM.py:73-96)."""
    step = centers_separation_factor * 2 * vortex_radius
    cx = np.arange(xmin + vortex_radius, xmax - vortex_radius + step, step)
    cy = np.arange(ymin + vortex_radius, ymax - vortex_radius + step, step)
    cxv, cyv = np.meshgrid(cx, cy, indexing="ij")
    sign = (-1.0) ** (np.arange(len(cx))[:, None] + np.arange(len(cy))[None, :])   # +,-,+ along x, flipped per column
    centres = np.stack([cxv.ravel(), cyv.ravel()], axis=1)
    return generate_free_vortices(len(centres), centres, vortex_radius, layerspervort, npervortlayer,
                                  (gamma * sign).ravel())


def generate_flowfield_turbulence(vortex_radius=0.2, vortex_density=0.8, gamma=0.5, xmin=-5, xmax=0, ymin=-3, ymax=2.5,
                                  layerspervort=2, npervortlayer=np.array([1, 5]), overlap=False, seed=None, rng=None):
    """Randomly placed clouds of random sign (LUDVM.py:99-130); without `overlap` centres are redrawn
    (up to 20000 times each) until they are at least two radii apart."""
    rng = rng if rng is not None else np.random.default_rng(seed)
    area = (xmax - xmin) * (ymax - ymin)
    nvorts = int(vortex_density * area / (np.pi * vortex_radius**2))
    gammas = gamma * rng.choice([-1, 1], nvorts)
    c = np.stack([rng.uniform(xmin, xmax, nvorts), rng.uniform(ymin, ymax, nvorts)], axis=1)
    for n in range(1, nvorts):
        tries = 0
        while True:
            if overlap or tries >= 20000:
                break
            if np.all(np.hypot(c[:n, 0] - c[n, 0], c[:n, 1] - c[n, 1]) >= 2 * vortex_radius):
                break
            c[n] = rng.uniform(xmin, xmax), rng.uniform(ymin, ymax)
            tries += 1
        if overlap:
            c[n] = rng.uniform(xmin, xmax), rng.uniform(ymin, ymax)
    return generate_free_vortices(nvorts, c, vortex_radius, layerspervort, npervortlayer, gammas)
