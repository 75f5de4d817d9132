"""Zero-camber stand-in for the PyPI package `airfoils` (absent from this image, no network).

TEST INFRASTRUCTURE ONLY -- used by oracle/gen_golden.py so that the *unmodified* reference class
(/root/reference/LUDVM.py) can run end-to-end here.  The reference touches the package only at
LUDVM.py:301-302, 312, 328-335, and for a symmetric NACA 00xx section everything it takes from it
reduces to a camber line that is identically zero (eta == 0 at LUDVM.py:335,340), independent of
the package's point distribution.  Anything else (cambered digits) raises, so no golden vector can
silently depend on arithmetic this stand-in does not reproduce.
"""
import numpy as np


class Airfoil:
    def __init__(self, n_points):
        xs = np.linspace(0.0, 1.0, n_points)
        self._x_upper = xs
        self._x_lower = xs.copy()
        self._y_upper = np.zeros(n_points)   # thickness is never read by the reference
        self._y_lower = np.zeros(n_points)
        self.all_points = np.zeros((2, 2 * n_points))

    @classmethod
    def NACA4(cls, naca_digits, n_points=200):
        if len(naca_digits) != 4 or naca_digits[:2] != "00":
            raise NotImplementedError("airfoils stand-in only covers symmetric NACA 00xx sections")
        return cls(n_points)

    def camber_line(self, x):
        return np.zeros_like(np.asarray(x, dtype=float))

    def camber_line_angle(self, x):
        return np.zeros_like(np.asarray(x, dtype=float))
