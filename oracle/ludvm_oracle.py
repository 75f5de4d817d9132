"""CPU restatement of the LUDVM hot path (float64 NumPy).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (ludvm_amd/) imports this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and there only as the
checker.  It restates the reference's algorithm (jcatalang/LUDVM, LUDVM.py) so that the parity
tests can run where the reference itself does not travel (the GPU box).

Pinned against the reference: tests/test_oracle_golden.py checks every function here against the
golden vectors under tests/golden/, which oracle/gen_golden.py produced by importing and running
the unmodified reference in the build container (kernel KATs, the config-1 run, the per-call
boundary trace, the flow field, and the Ramesh / free-vortex / alpha_m variants).
Unpinned: cambered NACA digits and .dat sections (the reference takes those from the PyPI package
`airfoils`, absent here; naca4_camber() below restates the published NACA 4-digit formula).

Citations are file:line into /root/reference/LUDVM.py.
"""
import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz  # same routine the reference calls as np.trapz


# --------------------------------------------------------------------------------------------
# the pair kernel
# --------------------------------------------------------------------------------------------
def induced_velocity(circulation, xw, zw, xp, zp, v_core, viscous=True, rows_per_chunk=None):
    """LUDVM.induced_velocity, LUDVM.py:549-570, arithmetic as written.

    float64 [Np, Nw] broadcast; the denominator is evaluated once for Ku and once for Kw (:565-566);
    the sum runs along the contiguous source axis (:569).  `viscous` is compared with `== True`
    (:562).  `rows_per_chunk` evaluates blocks of target rows: rows are independent and the row sum
    is row-local, so the result is bit-identical to the unchunked call.
    """
    circulation = np.asarray(circulation)
    xw = np.asarray(xw, dtype=float)
    zw = np.asarray(zw, dtype=float)
    xp = np.asarray(xp, dtype=float)
    zp = np.asarray(zp, dtype=float)
    vc = v_core if viscous == True else 0  # noqa: E712  (the reference's comparison)
    n_p = len(xp)
    u = np.empty(n_p)
    w = np.empty(n_p)
    step = n_p if not rows_per_chunk else int(rows_per_chunk)
    for a in range(0, n_p, max(step, 1)):
        b = min(n_p, a + step)
        x_dist = xp[a:b, None] - xw[None, :]
        z_dist = zp[a:b, None] - zw[None, :]
        ku = z_dist / (2 * np.pi * np.sqrt((x_dist**2 + z_dist**2) ** 2 + vc**4))
        kw = x_dist / (2 * np.pi * np.sqrt((x_dist**2 + z_dist**2) ** 2 + vc**4))
        u[a:b] = np.sum(circulation * ku, axis=1)
        w[a:b] = np.sum(-circulation * kw, axis=1)
    return u, w


def naca4_camber(digits, x):
    """Published NACA 4-digit mean line: m = d0/100, p = d1/10 (x in chord fractions)."""
    m = int(digits[0]) / 100.0
    p = int(digits[1]) / 10.0
    x = np.asarray(x, dtype=float)
    if m == 0.0 or p == 0.0:
        return np.zeros_like(x)
    fore = m / p**2 * (2 * p * x - x**2)
    aft = m / (1 - p) ** 2 * ((1 - 2 * p) + 2 * p * x - x**2)
    return np.where(x < p, fore, aft)


# --------------------------------------------------------------------------------------------
# the solver around it
# --------------------------------------------------------------------------------------------
class OracleLUDVM:
    """Restatement of class LUDVM (LUDVM.py:132-1298): same constructor keywords, same result
    attributes.  `run=False` builds geometry and kinematics only."""

    def __init__(self, t0=0, tf=12, dt=1.5e-2, chord=1, rho=1.225, Uinf=1, Npoints=80, Ncoeffs=30, LESPcrit=0.2,
                 Naca="0012", foil_filename=None, G=1, T=2, alpha_m=0, alpha_max=10, k=0.2 * np.pi, phi=90,
                 h_max=1, verbose=False, method="Faure", circulation_freevort=None, xy_freevort=None, run=True,
                 kernel=None):
        # parameters, LUDVM.py:237-260
        self.t0, self.tf, self.dt = t0, tf, dt
        self.chord, self.rho, self.Uinf = chord, rho, Uinf
        self.Npoints, self.Ncoeffs = Npoints, Ncoeffs
        self.piv = 0.25 * chord
        self.LESPcrit = LESPcrit
        self.maxerror, self.maxiter, self.epsilon = 1e-10, 50, 1e-4
        self.xgamma = 0.25
        self.method = method
        self.t = np.arange(t0, tf + dt, dt)
        self.nt = len(self.t)
        self.verbose = verbose
        self.dt_star = dt * Uinf / chord
        self.v_core = 1.3 * self.dt_star * chord
        self.alpha_m = alpha_m
        # free vortices, LUDVM.py:268-277
        if circulation_freevort is not None and xy_freevort is not None:
            self.n_freevort = len(circulation_freevort)
            self.circulation_freevort = circulation_freevort
            self.xy_freevort = xy_freevort
        else:
            self.n_freevort = 1
            self.circulation_freevort = np.array([0])
            self.xy_freevort = np.array([0, 0])[:, np.newaxis]
        # `kernel(circulation, xw, zw, xp, zp) -> (u, w)` lets a test substitute the pair sum
        self._kernel = kernel
        self.airfoil_generation(Naca)
        self.motion_sinusoidal(alpha_m=alpha_m, alpha_max=alpha_max, h_max=h_max, k=k, phi=phi, h0=0, x0=0)
        if run:
            self.time_loop()
            self.compute_coefficients()

    # -- L0 ----------------------------------------------------------------------------------
    def induced_velocity(self, circulation, xw, zw, xp, zp, viscous=True):
        if self._kernel is not None and viscous == True:  # noqa: E712
            return self._kernel(circulation, xw, zw, xp, zp)
        return induced_velocity(circulation, xw, zw, xp, zp, self.v_core, viscous)

    # -- L2 setup ----------------------------------------------------------------------------
    def airfoil_generation(self, Naca="0012"):
        """LUDVM.py:299-380 for NACA 4-digit sections, theta-uniform nodes."""
        c, n = self.chord, self.Npoints
        xa = np.linspace(0.0, 1.0, n)              # chordwise stations of the mean line
        etaa = c * naca4_camber(Naca, xa)          # :335
        theta = np.linspace(0, np.pi, n)           # :338
        x = c / 2 * (1 - np.cos(theta))            # :339
        eta = np.interp(x, c * xa, etaa)           # :340 (xa there is chord*0.5*(xupper+xlower))
        x_panel = x[:-1] + self.xgamma * (x[1:] - x[:-1])      # :345
        eta_panel = np.interp(x_panel, x, eta)                 # :346
        theta_panel = np.arccos(1 - 2 * x_panel / c)           # :347

        def slopes(f, s):
            # :350-372 -- one-sided at the ends, "(f[i+1]-f[i-1]) / (2*(s[i+1]-s[i-1]))" inside
            d = np.zeros(len(f))
            d[0] = (f[1] - f[0]) / (s[1] - s[0])
            d[-1] = (f[-1] - f[-2]) / (s[-1] - s[-2])
            d[1:-1] = (f[2:] - f[:-2]) / (2 * (s[2:] - s[:-2]))
            return d

        self.airfoil = {
            "x": x, "theta": theta, "eta": eta,
            "detadx": slopes(eta, x), "detadtheta": slopes(eta, theta),
            "x_panel": x_panel, "theta_panel": theta_panel, "eta_panel": eta_panel,
            "detadx_panel": slopes(eta_panel, x_panel), "detadtheta_panel": slopes(eta_panel, theta_panel),
        }

    def motion_sinusoidal(self, alpha_m=0, alpha_max=10, h_max=1, k=0.2 * np.pi, phi=90, h0=0, x0=0.25, motion="cos"):
        """LUDVM.py:382-457: pitch/heave tables and the rigid-body path of the nodes."""
        pi, U, t = np.pi, self.Uinf, self.t
        f = k * U / (2 * pi * self.chord)
        self.f = f
        alpha_m, alpha_max, phi = alpha_m * pi / 180, alpha_max * pi / 180, phi * pi / 180
        if motion == "cos":
            alpha = alpha_m + alpha_max * np.cos(2 * pi * f * t + phi)
            alpha_dot = -alpha_max * 2 * pi * f * np.sin(2 * pi * f * t + phi)
            h = h0 + h_max * np.cos(2 * pi * f * t)
            h_dot = -h_max * 2 * pi * f * np.sin(2 * pi * f * t)
        else:
            alpha = alpha_m + alpha_max * np.sin(2 * pi * f * t + phi)
            alpha_dot = alpha_max * 2 * pi * f * np.cos(2 * pi * f * t + phi)
            h = h0 + h_max * np.sin(2 * pi * f * t)
            h_dot = -h_max * 2 * pi * f * np.cos(2 * pi * f * t)
        xpiv = x0 - U * t
        self.alpha_e = alpha - np.arctan2(h_dot, U)
        ca, sa = np.cos(-alpha), np.sin(-alpha)
        path = np.zeros([self.nt, 2, self.Npoints])
        path[:, 0, 0] = xpiv - self.piv * ca                    # :435
        path[:, 1, 0] = h + self.piv * sa                       # :436
        xq, eq = self.airfoil["x"][1:], self.airfoil["eta"][1:]
        path[:, 0, 1:] = path[:, 0, :1] + ca[:, None] * xq - sa[:, None] * eq   # :441-442
        path[:, 1, 1:] = path[:, 1, :1] + sa[:, None] * xq + ca[:, None] * eq   # :443-444
        gpts = path[:, :, :-1] + self.xgamma * (path[:, :, 1:] - path[:, :, :-1])  # :447-448
        self.phi, self.h_max = phi, h_max
        self.alpha, self.alpha_dot = alpha, alpha_dot
        self.hpiv, self.h_dot = h, h_dot
        self.xpiv, self.x_dot = xpiv, -U * np.ones(self.nt)
        self.path = {"airfoil": path, "airfoil_gamma_points": gpts}

    # -- L1 ----------------------------------------------------------------------------------
    def _to_chord_frame(self, u1, w1, i):
        a = self.alpha[i]
        return u1 * np.cos(a) - w1 * np.sin(a), u1 * np.sin(a) + w1 * np.cos(a)   # :587-588

    def airfoil_downwash(self, circulation, xw, zw, i):
        """LUDVM.py:572-595."""
        a, ad, hd = self.alpha[i], self.alpha_dot[i], self.h_dot[i]
        xp, zp = self.path["airfoil_gamma_points"][i, 0, :], self.path["airfoil_gamma_points"][i, 1, :]
        u1, w1 = self.induced_velocity(circulation, xw, zw, xp, zp)
        u, w = self._to_chord_frame(u1, w1, i)
        af = self.airfoil
        return af["detadx_panel"] * (self.Uinf * np.cos(a) + hd * np.sin(a) + u - ad * af["eta_panel"]) \
            - self.Uinf * np.sin(a) - ad * (af["x_panel"] - self.piv) + hd * np.cos(a) - w

    # -- L3 ----------------------------------------------------------------------------------
    def _wake(self, i, n_tev, n_lev):
        """Gather circulation and row-i positions of TEV[:n_tev] ++ LEV[:n_lev] ++ FREE
        (the np.append chains at :689-691, :743-745, :1049-1051, :1095-1098)."""
        c, p = self.circulation, self.path
        g = np.concatenate([c["TEV"][:n_tev], c["LEV"][:n_lev], np.asarray(c["FREE"], dtype=float)])
        xw = np.concatenate([p["TEV"][i, 0, :n_tev], p["LEV"][i, 0, :n_lev], p["FREE"][i, 0, :]])
        zw = np.concatenate([p["TEV"][i, 1, :n_tev], p["LEV"][i, 1, :n_lev], p["FREE"][i, 1, :]])
        return g, xw, zw

    def _kelvin(self, lead, n_tev, n_lev):
        """lead + sum(TEV) + sum(LEV) + sum(FREE) - IC, associated left to right as the reference
        writes it (:698-699, :758-760, :946-948): the run is chaotic, so even the rounding of this
        sum has to match for the late-time trajectories to agree."""
        c = self.circulation
        return lead + np.sum(c["TEV"][:n_tev]) + np.sum(c["LEV"][:n_lev]) + np.sum(c["FREE"]) - c["IC"]

    def _unit_vortex_term(self, x0, z0, i):
        """T2 / T3 of the Faure solve (:749-754, :924-934): chord-normal influence of a unit vortex."""
        xa, za = self.path["airfoil_gamma_points"][i, 0, :], self.path["airfoil_gamma_points"][i, 1, :]
        u1, w1 = self.induced_velocity(np.array([1]), np.array([x0]), np.array([z0]), xa, za)
        ut, un = self._to_chord_frame(u1, w1, i)
        return self.airfoil["detadx_panel"] * ut - un

    def _fourier_from_W(self, W, i, first):
        th, U = self.airfoil["theta_panel"], self.Uinf
        for n in range(first, self.Ncoeffs):
            self.fourier[i, 0, n] = 2 / np.pi * _trapz(W / U * np.cos(n * th), th)

    def _a0a1(self, W):
        th, U = self.airfoil["theta_panel"], self.Uinf
        return -1 / np.pi * _trapz(W / U, th), 2 / np.pi * _trapz(W / U * np.cos(th), th)

    def time_loop(self, print_dt=50, BCcheck=False):
        """LUDVM.py:597-1171."""
        pi, U, c, rho, dt = np.pi, self.Uinf, self.chord, self.rho, self.dt
        th, thp = self.airfoil["theta"], self.airfoil["theta_panel"]
        nt, nv, nf, npan = self.nt, self.nt - 1, self.n_freevort, self.Npoints - 1
        lesp_crit = self.LESPcrit
        P = self.path
        P["TEV"] = np.zeros([nt, 2, nv])
        P["LEV"] = np.zeros([nt, 2, nv])
        P["FREE"] = np.zeros([nt, 2, nf])
        P["FREE"][0, :, :] = self.xy_freevort
        C = self.circulation = {"TEV": np.zeros(nv), "LEV": np.zeros(nv), "FREE": self.circulation_freevort,
                                "bound": np.zeros(nv), "airfoil": np.zeros([nv, npan]),
                                "gamma_airfoil": np.zeros([nv, npan]), "Gamma_airfoil": np.zeros([nv, npan])}
        self.BC = np.zeros([nv, self.Npoints])
        self.dp = np.zeros([nt, npan])
        self.Fn, self.Fs, self.L, self.D, self.T, self.M = (np.zeros(nt) for _ in range(6))
        self.fourier = np.zeros([nt, 2, self.Ncoeffs])
        self.LESP, self.LESP_prev = np.zeros(nt), np.zeros(nt)
        A0, A1 = np.sin(self.alpha_m), 0                       # :645 (alpha_m in degrees, as written)
        self.fourier[0, 0, :2] = A0, A1
        C["IC"] = np.sum(C["FREE"]) + U * c * pi * (A0 + A1 / 2)   # :646-649
        itev = ilev = 0
        LEV_shed = -1 * np.ones(nt)

        for i in range(1, nt):
            # carry the wake over (:664-666) and place the new TEV (:672-681)
            P["TEV"][i, :, :itev] = P["TEV"][i - 1, :, :itev]
            P["LEV"][i, :, :ilev] = P["LEV"][i - 1, :, :ilev]
            P["FREE"][i] = P["FREE"][i - 1]
            te = P["airfoil"][i, :, -1]
            if itev == 0:
                P["TEV"][i, :, 0] = P["airfoil"][0, :, -1] + [0.5 * U * dt, 0]
            else:
                P["TEV"][i, :, itev] = te + 1 / 3 * (P["TEV"][i, :, itev - 1] - te)

            if self.method == "Ramesh":
                self._ramesh_tev(i, itev, ilev)
            else:
                # Faure closed form (:741-773)
                T1 = self.airfoil_downwash(*self._wake(i, itev, ilev), i)
                T2 = self._unit_vortex_term(P["TEV"][i, 0, itev], P["TEV"][i, 1, itev], i)
                I1 = _trapz(T1 * (np.cos(thp) - 1), thp)
                I2 = _trapz(T2 * (np.cos(thp) - 1), thp)
                C["TEV"][itev] = -self._kelvin(I1, itev, ilev) / (1 + I2)
                C["bound"][itev] = I1 + C["TEV"][itev] * I2
                W = T1 + C["TEV"][itev] * T2
                self.fourier[i, 0, 0] = -1 / pi * _trapz(W / U, thp)
                self._fourier_from_W(W, i, 1)
                self.fourier[i, 1, :] = (self.fourier[i, 0, :] - self.fourier[i - 1, 0, :]) / dt
            self.LESP_prev[itev] = self.fourier[i, 0, 0]

            if abs(self.fourier[i, 0, 0]) >= abs(lesp_crit):     # :781
                LEV_shed[i] = ilev
                le = P["airfoil"][i, :, 0]
                if ilev > 0 and LEV_shed[i - 1] != -1:           # :788-800
                    P["LEV"][i, :, ilev] = le + 1 / 3 * (P["LEV"][i, :, ilev - 1] - le)
                else:
                    P["LEV"][i, :, ilev] = le
                lesp_crit = -abs(lesp_crit) if self.fourier[i, 0, 0] < 0 else abs(lesp_crit)   # :802-805
                if self.method == "Ramesh":
                    self._ramesh_tev_lev(i, itev, ilev, lesp_crit)
                else:
                    # 2x2 Faure system (:916-961); derivatives keep their pre-LEV values (:963-966)
                    T1 = self.airfoil_downwash(*self._wake(i, itev, ilev), i)
                    T2 = self._unit_vortex_term(P["TEV"][i, 0, itev], P["TEV"][i, 1, itev], i)
                    T3 = self._unit_vortex_term(P["LEV"][i, 0, ilev], P["LEV"][i, 1, ilev], i)
                    cm1 = np.cos(thp) - 1
                    I1, I2, I3 = _trapz(T1 * cm1, thp), _trapz(T2 * cm1, thp), _trapz(T3 * cm1, thp)
                    J1, J2, J3 = (-1 / np.pi * _trapz(T, thp) for T in (T1, T2, T3))
                    A = np.array([[1 + I2, 1 + I3], [J2, J3]])
                    b = np.array([-self._kelvin(I1, itev, ilev), lesp_crit - J1])
                    C["TEV"][itev], C["LEV"][ilev] = np.linalg.solve(A, b)
                    C["bound"][itev] = I1 + C["TEV"][itev] * I2 + C["LEV"][ilev] * I3
                    W = T1 + C["TEV"][itev] * T2 + C["LEV"][ilev] * T3
                    self.fourier[i, 0, 0] = J1 + C["TEV"][itev] * J2 + C["LEV"][ilev] * J3
                    self._fourier_from_W(W, i, 1)
            self.LESP[itev] = self.fourier[i, 0, 0]

            # bound vorticity per panel (:987-1010); term2 accumulates n = 1, 2, ... in that order
            A0, A0d = self.fourier[i, :, 0]
            A1, A1d = self.fourier[i, :, 1]
            A2, A2d = self.fourier[i, :, 2]
            _, A3d = self.fourier[i, :, 3]
            term2 = np.zeros(npan)
            for n in range(1, self.Ncoeffs):
                term2 = self.fourier[i, 0, n] * np.sin(n * thp) + term2
            gamma = 2 * U * (A0 * (1 + np.cos(thp)) / np.sin(thp) + term2)
            dGamma = gamma * c / 2 * np.sin(thp) * (th[1:] - th[:-1])
            C["airfoil"][itev], C["gamma_airfoil"][itev] = dGamma, gamma
            for j in range(npan):
                C["Gamma_airfoil"][itev, j] = np.sum(dGamma[: j + 1])

            # loads (:1035-1090)
            a, hd = self.alpha[i], self.h_dot[i]
            x_gamma = self.airfoil["x_panel"]
            gw, xw, zw = self._wake(i, itev + 1, ilev + 1)
            xg, zg = P["airfoil_gamma_points"][i, 0, :], P["airfoil_gamma_points"][i, 1, :]
            u1, w1 = self.induced_velocity(gw, xw, zw, np.array(xg), np.array(zg))
            u, _ = self._to_chord_frame(u1, w1, i)
            Ueff = U * np.cos(a) + hd * np.sin(a)
            self.Fn[i] = rho * pi * c * U * (Ueff * (A0 + 0.5 * A1) + c * (3 / 4 * A0d + 1 / 4 * A1d + 1 / 8 * A2d)) \
                + rho * _trapz(u * gamma, x_gamma)
            self.Fs[i] = rho * pi * c * U**2 * A0**2
            self.L[i] = self.Fn[i] * np.cos(a) + self.Fs[i] * np.sin(a)
            self.D[i] = self.Fn[i] * np.sin(a) - self.Fs[i] * np.cos(a)
            self.T[i] = -self.D[i]
            self.M[i] = self.piv * self.Fn[i] - rho * pi * c**2 * U * (
                Ueff * (1 / 4 * A0 + 1 / 4 * A1 - 1 / 8 * A2)
                + c * (7 / 16 * A0d + 3 / 16 * A1d + 1 / 16 * A2d - 1 / 64 * A3d)) \
                - rho * _trapz(u * gamma * x_gamma, x_gamma)

            # wake roll-up (:1095-1127): three target slices, identical sources, explicit Euler
            for key, cnt in (("TEV", itev + 1), ("LEV", ilev + 1), ("FREE", nf)):
                xp, zp = P[key][i, 0, :cnt], P[key][i, 1, :cnt]
                uw, ww = self.induced_velocity(gw, xw, zw, xp, zp)
                uf, wf = self.induced_velocity(dGamma, xg, zg, xp, zp, viscous=True)
                P[key][i, 0, :cnt] = xp + dt * (uw + uf)
                P[key][i, 1, :cnt] = zp + dt * (ww + wf)

            self.ilev, self.itev, self.LEV_shed = ilev, itev, LEV_shed
            if LEV_shed[i] != -1:
                ilev += 1
            itev += 1

    # Newton variants (:683-739, :807-914) -------------------------------------------------------
    def _ramesh_eval(self, i, itev, ilev):
        W = self.airfoil_downwash(*self._wake(i, itev + 1, ilev + 1), i)
        A0, A1 = self._a0a1(W)
        bound = self.Uinf * self.chord * np.pi * (A0 + A1 / 2)
        return W, A0, A1, bound, self._kelvin(bound, itev + 1, ilev + 1)

    def _ramesh_tev(self, i, itev, ilev):
        C, eps = self.circulation, self.epsilon
        f, niter, g = 1, 1, -1
        while abs(f) > self.maxerror and niter < self.maxiter:
            C["TEV"][itev] = g
            f = self._ramesh_eval(i, itev, ilev)[4]
            C["TEV"][itev] = g + eps
            fd = self._ramesh_eval(i, itev, ilev)[4]
            g = g - f / ((fd - f) / eps)
            C["TEV"][itev] = g
            niter += 1
        W, A0, A1, bound, _ = self._ramesh_eval(i, itev, ilev)
        self.fourier[i, 0, :2] = A0, A1
        C["bound"][itev] = bound
        self._fourier_from_W(W, i, 2)
        self.fourier[i, 1, :] = (self.fourier[i, 0, :] - self.fourier[i - 1, 0, :]) / self.dt

    def _ramesh_tev_lev(self, i, itev, ilev, lesp_crit):
        C, eps = self.circulation, self.epsilon
        g_lev = g_tev = C["TEV"][itev]
        f1 = f2 = 0.1
        niter = 1
        while (abs(f1) > self.maxerror or abs(f2) > self.maxerror) and niter < self.maxiter:
            def residuals(gt, gl):
                C["TEV"][itev], C["LEV"][ilev] = gt, gl
                _, A0, _, bound, kel = self._ramesh_eval(i, itev, ilev)
                return kel, lesp_crit - A0, bound
            f1, f2, cbound = residuals(g_tev, g_lev)
            f1t, f2t, _ = residuals(g_tev + eps, g_lev)
            f1l, f2l, _ = residuals(g_tev, g_lev + eps)
            J = np.array([[(f1l - f1) / eps, (f1t - f1) / eps], [(f2l - f2) / eps, (f2t - f2) / eps]])
            g_lev, g_tev = np.array([g_lev, g_tev]) - np.linalg.solve(J, np.array([f1, f2]))
            C["TEV"][itev], C["LEV"][ilev], C["bound"][itev] = g_tev, g_lev, cbound
            niter += 1
        W, A0, A1, bound, _ = self._ramesh_eval(i, itev, ilev)
        self.fourier[i, 0, :2] = A0, A1
        C["bound"][itev] = bound
        self._fourier_from_W(W, i, 2)

    def compute_coefficients(self):
        """LUDVM.py:1173-1184."""
        q = 0.5 * self.rho * self.Uinf**2
        qc = q * self.chord
        self.Cp = self.dp / q
        self.Cn, self.Cs = self.Fn / qc, self.Fs / qc
        self.Cl, self.Cd, self.Ct = self.L / qc, self.D / qc, self.T / qc
        self.Cm = self.M / (qc * self.chord)

    # -- L3' -----------------------------------------------------------------------------------
    def flowfield_sources(self, s):
        """Source gather of LUDVM.flowfield for time step s (:1202-1215), index quirks included:
        TEV/LEV positions come from row s-1 but carry slots [:s+1] / [:ilev+1]; FREE from row s;
        LEV_shed[s] == -1 drops every LEV."""
        C, P = self.circulation, self.path
        if s == 0:
            return (np.asarray(C["FREE"], float), P["FREE"][0, 0, :], P["FREE"][0, 1, :]), None
        ilev = int(self.LEV_shed[s])
        g = np.concatenate([C["TEV"][: s + 1], C["LEV"][: ilev + 1], np.asarray(C["FREE"], float)])
        xw = np.concatenate([P["TEV"][s - 1, 0, : s + 1], P["LEV"][s - 1, 0, : ilev + 1], P["FREE"][s, 0, :]])
        zw = np.concatenate([P["TEV"][s - 1, 1, : s + 1], P["LEV"][s - 1, 1, : ilev + 1], P["FREE"][s, 1, :]])
        foil = (C["airfoil"][s - 1, :], P["airfoil_gamma_points"][s - 1, 0, :], P["airfoil_gamma_points"][s - 1, 1, :])
        return (g, xw, zw), foil

    def flowfield(self, xmin=-10, xmax=0, zmin=-4, zmax=4, dr=0.02, tsteps=(0, 1, 2)):
        """LUDVM.py:1186-1298."""
        x1, z1 = np.arange(xmin, xmax, dr), np.arange(zmin, zmax, dr)
        x, z = np.meshgrid(x1, z1, indexing="ij")
        xp, zp = np.ravel(x), np.ravel(z)
        u = np.zeros([len(tsteps), *x.shape])
        w = np.zeros_like(u)
        for ii, s in enumerate(tsteps):
            wake, foil = self.flowfield_sources(s)
            uu, ww = self.induced_velocity(*wake, xp, zp)
            if foil is not None:
                uf, wf = self.induced_velocity(*foil, xp, zp)
                uu, ww = uu + uf, ww + wf
            u[ii], w[ii] = uu.reshape(x.shape), ww.reshape(x.shape)
        self.x_ff, self.z_ff, self.u_ff, self.w_ff = x, z, u, w
        self.ome_ff = vorticity(u, w, x, z)


def vorticity(u, w, x, z):
    """ome = dw/dx - du/dz on the mesh (LUDVM.py:1224-1292): centred inside, one-sided on edges and
    corners, mesh spacings taken from x and z themselves.  u, w are [nsteps, nx, nz]."""
    nx, nz = x.shape
    ip = np.minimum(np.arange(nx) + 1, nx - 1)
    im = np.maximum(np.arange(nx) - 1, 0)
    jp = np.minimum(np.arange(nz) + 1, nz - 1)
    jm = np.maximum(np.arange(nz) - 1, 0)
    dx = (x[ip, :] - x[im, :])[None]
    dz = (z[:, jp] - z[:, jm])[None]
    return (w[:, ip, :] - w[:, im, :]) / dx - (u[:, :, jp] - u[:, :, jm]) / dz
