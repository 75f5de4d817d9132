"""Drop-in `LUDVM` class: the reference's constructor keywords, methods and result attributes
(jcatalang/LUDVM, LUDVM.py:132-1372), with the O(N^2) Vatistas-core Biot-Savart sums evaluated by
the gfx950 engine (libludvm_hip.so through ludvm_amd.engine.Engine).

What runs where
  * device: every induced-velocity evaluation -- wake -> chord points, unit new TEV/LEV -> chord
    points, wake+foil -> wake (fused with the explicit-Euler update), wake+foil -> flow-field grid,
    the vorticity stencil;  the wake stays resident on the device across time steps;
  * host (float64 NumPy, O(Npanels * Ncoeffs) per step): kinematics, vortex placement, the Gamma_TEV /
    Gamma_LEV solve, Fourier projection A0..A_N, bound-vorticity reconstruction, loads.  These are
    written as small matrix products (trapezoid weights, cos/sin tables) instead of the reference's
    interpreted loops (LUDVM.py:770-773, :994-1010).

There is no CPU implementation of the pair sum in this package: without the HIP library and an
MI355X the constructor raises.

Citations are file:line into the reference's LUDVM.py.
"""
import timeit

import numpy as np

from .engine import Engine

__all__ = ["LUDVM", "SparseHistory"]

# The reference's dense trajectory arrays path['TEV'|'LEV'] are [nt, 2, nt-1] float64 (LUDVM.py:615-616): 32 nt^2 bytes.
# 'auto' keeps them -- and float64 pair sums, which cost nothing at those wake sizes -- while they fit this budget
# (nt <= 8191: every case the reference itself can run in a few minutes); beyond it rows are kept at
# snapshot_steps only and the wake-on-wake sums run in fp32 with local origins.
_DENSE_HISTORY_BUDGET_BYTES = 2 << 30


def _dense_history_fits(nt):
    return 32 * nt * nt <= _DENSE_HISTORY_BUDGET_BYTES


def naca4_mean_line(digits, x):
    """NACA 4-digit mean line (m = d0/100 at p = d1/10), x in chord fractions."""
    m, p = int(digits[0]) / 100.0, int(digits[1]) / 10.0
    x = np.asarray(x, dtype=float)
    if m == 0.0 or p == 0.0:
        return np.zeros_like(x)
    return np.where(x < p, m / p**2 * (2 * p * x - x**2), m / (1 - p) ** 2 * ((1 - 2 * p) + 2 * p * x - x**2))


class SparseHistory:
    """Trajectory rows kept only for selected time steps (the reference's dense [nt, 2, n] arrays are
    O(nt^2): 2 x 40 GB at dt = 1e-3, t in [0, 50]).  Indexing follows the dense arrays for the rows
    that exist: h[i] -> [2, n_i], h[i, 0, :k] -> view; other rows raise KeyError."""

    def __init__(self, nt):
        self.nt = nt
        self.rows = {}

    def store(self, i, row):
        self.rows[int(i)] = row

    def steps(self):
        return sorted(self.rows)

    def __contains__(self, i):
        return int(i) in self.rows

    def __getitem__(self, key):
        if isinstance(key, tuple):
            i, rest = key[0], key[1:]
        else:
            i, rest = key, ()
        i = int(i)
        if i < 0:
            i += self.nt
        if i not in self.rows:
            raise KeyError(f"time step {i} was not recorded (history='sparse'; recorded: {self.steps()[:8]}...)")
        row = self.rows[i]
        return row[rest] if rest else row


class LUDVM:
    """LESP-modulated unsteady discrete vortex method (see the reference's class docstring,
    LUDVM.py:133-229) on an MI355X.

    Same positional/keyword parameters as the reference constructor (LUDVM.py:231-236) and, like it,
    the whole simulation runs inside the constructor.  Keyword-only extras:
      engine     an existing ludvm_amd.engine.Engine to use (default: a new one on `device`)
      device     HIP device ordinal
      precision  arithmetic of the wake-on-wake pair sums: 'f32' (fp32 on local-origin positions: offsets from the
                 origin of each 256-vortex block of the wake), 'f32x2' (hi+lo fp32 positions), 'f64', or 'auto'
                 (default): 'f64' for runs short enough to keep the reference's dense history (nt <= 8191, wake
                 <= 16 000 vortices: fp64 costs little there and the reference's numbers are reproduced to ~1e-5
                 over the whole README run), 'f32' beyond.  The Npanels-target sums always run in fp64
      history    'full' (dense path arrays as in the reference), 'sparse' (rows only at
                 snapshot_steps + last step) or 'auto' (full while the arrays fit 2 GiB, nt <= 8191)
      snapshot_steps  iterable of time-step indices to record when history is sparse
      run        False builds geometry and kinematics only
      checkpoint_every, checkpoint_path   write an .npz checkpoint every so many steps (0 = never);
                 `LUDVM.resume(path)` continues such a run (the reference has no checkpointing)
      march      True (default): stretches of time steps whose history row is not recorded run as a
                 device-resident march (Gamma solve on the GPU, no host round trip per step);
                 False: one device round trip per step throughout
      devices    several GPUs of this node in ONE process, no launcher: an int G (devices 0 .. G-1) or a list of ordinals.  One host
                 thread, one engine and one replica per device, the library's own RCCL communicator over them (ncclCommInitAll);
                 the object returned is a front whose attributes are replica 0's and whose methods run on all replicas
                 (ludvm_amd/multi.py; `close()` ends the threads).  One device: an ordinary run on it
      distributed  None (default): one GPU.  'rccl': the simulation is shared by the processes of one launch (one per GPU;
                 rank and world from the launcher's environment), connected by the library's own RCCL communicator --
                 no torch.distributed (ludvm_amd/comm.py).  True or a torch.distributed process group: the same through
                 torch's collectives (ludvm_amd/distributed.py).  Either way every rank constructs the same object and
                 calls the same methods; flowfield shards the grid rows, induced_velocity the targets, time_loop the
                 roll-up's unordered pairs (one integer all-reduce per step over RCCL), and every rank ends up with the
                 reference's full result arrays
    """

    def __new__(cls, *args, devices=None, **kwargs):
        # devices=[...] with more than one GPU: ONE process, one host thread and one replica per device, the library's own
        # communicator over them (ludvm_amd/multi.py) -- the object returned is that front, not an instance of this class
        if devices is not None and cls is LUDVM:
            from .multi import MultiDeviceLUDVM, normalise_devices
            devs = normalise_devices(devices)
            if len(devs) > 1:
                return MultiDeviceLUDVM(args, kwargs, devs)
        return super().__new__(cls)

    def __init__(self, t0=0, tf=12, dt=1.5e-2, chord=1, rho=1.225, Uinf=1,
                 Npoints=80, Ncoeffs=30, LESPcrit=0.2, Naca='0012',
                 foil_filename=None, G=1, T=2, alpha_m=0,
                 alpha_max=10, k=0.2 * np.pi, phi=90, h_max=1,
                 verbose=True, method='Faure',
                 circulation_freevort=None, xy_freevort=None, *,
                 engine=None, device=0, precision='auto', history='auto', snapshot_steps=(), run=True,
                 checkpoint_every=0, checkpoint_path=None, march=True, distributed=None, devices=None):
        if devices is not None:             # (one device: an ordinary single-GPU run on it)
            from .multi import normalise_devices
            device = normalise_devices(devices)[0]
        self._ctor = dict(t0=t0, tf=tf, dt=dt, chord=chord, rho=rho, Uinf=Uinf, Npoints=Npoints, Ncoeffs=Ncoeffs,
                          LESPcrit=LESPcrit, Naca=Naca, foil_filename=foil_filename, G=G, T=T, alpha_m=alpha_m,
                          alpha_max=alpha_max, k=k, phi=phi, h_max=h_max, method=method, precision=precision,
                          history=history, snapshot_steps=sorted(int(s) for s in snapshot_steps))
        # parameters (LUDVM.py:237-263)
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
        self.ilev2 = 0
        self.alpha_m = alpha_m
        # free vortices (LUDVM.py:268-277): default is one zero-strength vortex at the origin
        if circulation_freevort is not None and xy_freevort is not None:
            self.n_freevort = len(circulation_freevort)
            self.circulation_freevort = circulation_freevort
            self.xy_freevort = xy_freevort
        else:
            self.n_freevort = 1
            self.circulation_freevort = np.array([0])
            self.xy_freevort = np.array([0, 0])[:, np.newaxis]

        if precision not in ('auto', 'f32', 'f32x2', 'f64'):
            raise ValueError("precision must be 'auto', 'f32', 'f32x2' or 'f64'")
        if history not in ('auto', 'full', 'sparse'):
            raise ValueError("history must be 'auto', 'full' or 'sparse'")
        fits = _dense_history_fits(self.nt)
        if (precision == 'auto' or history == 'auto') and not fits and verbose:
            import warnings
            warnings.warn(f"LUDVM: nt = {self.nt}: the reference's dense trajectory history would need "
                          f"{32 * self.nt * self.nt / 2**30:.1f} GiB; 'auto' keeps rows at snapshot_steps only "
                          "(history='sparse') and runs the wake-on-wake sums in fp32 with local origins "
                          "(precision='f32').  Pass history='full' / precision='f64' to override.", stacklevel=2)
        if precision == 'auto':
            # an fp32 rounding difference grows ~10x per 12 steps once the wake rolls up: in fp32
            # the README case ends 0.1 away from the reference on Cl, in fp64 4e-5 -- at the same speed for small wakes.
            # Long runs (config 2) are chaotic beyond ~1000 steps in any arithmetic: 'f32' (local origins) passes the
            # whole-horizon statistical test against float64 runs (tests/test_gpu_cfg2_stats.py)
            precision = 'f64' if fits else 'f32'
        self.precision = precision
        self.history = ('full' if fits else 'sparse') if history == 'auto' else history
        self.snapshot_steps = {int(s) for s in snapshot_steps}
        self.checkpoint_every, self.checkpoint_path = int(checkpoint_every), checkpoint_path
        self.march = bool(march)
        if self.checkpoint_every and not checkpoint_path:
            raise ValueError("checkpoint_every needs a checkpoint_path")
        self.engine = engine if engine is not None else Engine(device)  # raises without the HIP library / GPU
        self._shard = None
        if isinstance(distributed, str) and distributed == 'rccl':
            from .comm import LibraryGroup          # the library's own communicator: no torch.distributed
            self._shard = LibraryGroup(self.engine)
        elif distributed is not None and distributed is not False:
            if hasattr(distributed, 'gather_blocks'):          # a ShardGroup or LibraryGroup
                self._shard = distributed
            else:
                from .distributed import ShardGroup
                self._shard = ShardGroup(None if distributed is True else distributed)

        self.start_time = timeit.default_timer()
        if Naca is not None:
            self.airfoil_generation(Naca=Naca)
        else:
            self.airfoil_generation(Naca=None, filename=foil_filename)
        self.motion_sinusoidal(alpha_m=alpha_m, alpha_max=alpha_max, h_max=h_max, k=k, phi=phi, h0=0, x0=0,
                               motion='cos')
        if run:
            self.time_loop()
            self.compute_coefficients()
            if self.verbose:
                print('Elapsed time:', timeit.default_timer() - self.start_time)

    # ------------------------------------------------------------------------------------------
    # geometry and kinematics (host, run once)
    # ------------------------------------------------------------------------------------------
    def airfoil_generation(self, Naca='0012', filename=None, Npoints=None, uniform_spacing='theta'):
        """Mean line on theta-uniform nodes, panel quarter points and slopes (LUDVM.py:299-380).
        NACA 4-digit mean lines come from the published formula (the reference takes them from the
        PyPI package `airfoils`); Selig-format .dat files are averaged upper/lower as at :316-324."""
        if Npoints is None:
            Npoints = self.Npoints
        c = self.chord
        from_file = Naca is None and filename is not None
        if from_file or str(Naca)[:2] != '00':
            # (symmetric NACA 00xx sections -- every BASELINE config -- have a zero mean line whatever its source)
            import warnings
            warnings.warn("LUDVM: the mean line of a cambered NACA section / a .dat file is restated from the published "
                          "NACA 4-digit formula (or the averaged surfaces of the file); the reference takes it from the PyPI "
                          "package `airfoils` (LUDVM.py:301-335: mean of the interpolated surfaces), which is not available "
                          "to this build -- parity with the reference is UNPINNED for this section (symmetric 00xx sections "
                          "are pinned).", RuntimeWarning, stacklevel=3)
        if from_file:
            xa, etaa = self._mean_line_from_dat(filename, Npoints)
        else:
            xs = np.linspace(0.0, 1.0, Npoints)
            xa, etaa = c * xs, c * naca4_mean_line(Naca, xs)
        if uniform_spacing == 'theta':
            theta = np.linspace(0, np.pi, self.Npoints)
            x = c / 2 * (1 - np.cos(theta))
            eta = np.interp(x, xa, etaa)
        else:
            x, eta = xa, etaa
            theta = np.arccos(1 - 2 * x / c)
        x_panel = x[:-1] + self.xgamma * (x[1:] - x[:-1])
        eta_panel = np.interp(x_panel, x, eta)
        theta_panel = np.arccos(1 - 2 * x_panel / c)

        def slope(f, s):  # one-sided at the ends; the reference's interior form divides by 2*(s[i+1]-s[i-1])
            d = np.empty(len(f))
            d[0] = (f[1] - f[0]) / (s[1] - s[0])
            d[-1] = (f[-1] - f[-2]) / (s[-1] - s[-2])
            d[1:-1] = (f[2:] - f[:-2]) / (2 * (s[2:] - s[:-2]))
            return d

        self.Npoints = len(eta)
        self.airfoil = {'x': x, 'theta': theta, 'eta': eta,
                        'detadx': slope(eta, x), 'detadtheta': slope(eta, theta),
                        'x_panel': x_panel, 'theta_panel': theta_panel, 'eta_panel': eta_panel,
                        'detadx_panel': slope(eta_panel, x_panel),
                        'detadtheta_panel': slope(eta_panel, theta_panel)}
        return None

    def _mean_line_from_dat(self, filename, Npoints):
        from scipy.interpolate import interp1d
        pts = []
        with open(filename) as fh:
            for line in fh:
                tok = line.split()
                if len(tok) == 2:
                    try:
                        pts.append((float(tok[0]), float(tok[1])))
                    except ValueError:
                        pass
        pts = np.array([p for p in pts if -0.01 <= p[0] <= 1.01])
        ile = int(np.argmin(pts[:, 0]))
        upper, lower = pts[: ile + 1][::-1], pts[ile:]
        n = int(np.floor(Npoints / 2))
        xs = np.linspace(0, 1, n)
        yu = interp1d(upper[:, 0], upper[:, 1], kind='cubic', bounds_error=False, fill_value='extrapolate')(xs)
        yl = interp1d(lower[:, 0], lower[:, 1], kind='cubic', bounds_error=False, fill_value='extrapolate')(xs)
        return self.chord * xs, self.chord * 0.5 * (yu + yl)

    def _rigid_body_path(self, xpiv, hpiv, alpha):
        """Node positions under pitch about the pivot (LUDVM.py:431-448)."""
        ca, sa = np.cos(-alpha)[:, None], np.sin(-alpha)[:, None]
        path = np.zeros([self.nt, 2, self.Npoints])
        path[:, 0, 0] = xpiv - self.piv * ca[:, 0]
        path[:, 1, 0] = hpiv + self.piv * sa[:, 0]
        xq, eq = self.airfoil['x'][1:], self.airfoil['eta'][1:]
        path[:, 0, 1:] = path[:, 0, :1] + ca * xq - sa * eq
        path[:, 1, 1:] = path[:, 1, :1] + sa * xq + ca * eq
        gpts = path[:, :, :-1] + self.xgamma * (path[:, :, 1:] - path[:, :, :-1])
        self.path = {'airfoil': path, 'airfoil_gamma_points': gpts}

    def motion_sinusoidal(self, alpha_m=0, alpha_max=10, h_max=1, k=0.2 * np.pi, phi=90, h0=0, x0=0.25,
                          motion='cos'):
        """h(t) = h0 + h_max cos(2 pi f t), alpha(t) = alpha_m + alpha_max cos(2 pi f t + phi),
        x(t) = x0 - Uinf t  (LUDVM.py:382-457); angles in degrees."""
        pi, U, t = np.pi, self.Uinf, self.t
        f = k * U / (2 * pi * self.chord)
        self.f = f
        alpha_m, alpha_max, phi = alpha_m * pi / 180, alpha_max * pi / 180, phi * pi / 180
        wt = 2 * pi * f * t
        if motion == 'cos':
            alpha, alpha_dot = alpha_m + alpha_max * np.cos(wt + phi), -alpha_max * 2 * pi * f * np.sin(wt + phi)
            h, h_dot = h0 + h_max * np.cos(wt), -h_max * 2 * pi * f * np.sin(wt)
        elif motion == 'sin':
            alpha, alpha_dot = alpha_m + alpha_max * np.sin(wt + phi), alpha_max * 2 * pi * f * np.cos(wt + phi)
            h, h_dot = h0 + h_max * np.sin(wt), -h_max * 2 * pi * f * np.cos(wt)
        else:
            raise ValueError("motion must be 'cos' or 'sin'")
        x = x0 - U * t
        self._rigid_body_path(x, h, alpha)
        self.phi, self.h_max = phi, h_max
        self.alpha, self.alpha_dot = alpha, alpha_dot
        self.alpha_e = alpha - np.arctan2(h_dot, U)
        self.hpiv, self.h_dot = h, h_dot
        self.xpiv, self.x_dot = x, -U * np.ones(self.nt)
        return None

    # ------------------------------------------------------------------------------------------
    # the hot path
    # ------------------------------------------------------------------------------------------
    def induced_velocity(self, circulation, xw, zw, xp, zp, viscous=True):
        """Velocity (u, w) induced at points (xp, zp) by vortices (circulation, xw, zw): Vatistas core
        v_core when `viscous == True`, point vortices otherwise (LUDVM.py:549-570).  Any array-likes
        in, two new float64 arrays out; evaluated on the GPU in `self.precision`."""
        v_core = self.v_core if viscous == True else 0  # noqa: E712  (the reference's comparison, :562)
        sh = self._shard
        if sh is not None and sh.world > 1 and len(xp) >= sh.min_targets and len(xp) * len(xw) >= getattr(sh, 'min_pairs', 0):
            # targets in contiguous blocks over the ranks (sources replicated), one all-gather of the (u, w) blocks -- for calls
            # large enough to repay it (many targets AND enough pairs: comm.py, MIN_TARGETS / MIN_PAIRS)
            xt, zt = np.asarray(xp, dtype=float).reshape(-1), np.asarray(zp, dtype=float).reshape(-1)
            lo, hi, _ = sh.block(len(xt))
            ul, wl = self.engine.induce(circulation, xw, zw, xt[lo:hi], zt[lo:hi], v_core, precision=self.precision)
            uw = sh.gather_blocks(np.stack([ul, wl], axis=1), len(xt))
            return np.ascontiguousarray(uw[:, 0]), np.ascontiguousarray(uw[:, 1])
        return self.engine.induce(circulation, xw, zw, xp, zp, v_core, precision=self.precision)

    def _chord_frame(self, u1, w1, i):
        a = self.alpha[i]
        return u1 * np.cos(a) - w1 * np.sin(a), u1 * np.sin(a) + w1 * np.cos(a)

    def _downwash_from(self, u1, w1, i):
        """W(x, t) from global-frame induced velocities at the bound-vortex points (LUDVM.py:586-593)."""
        a, ad, hd = self.alpha[i], self.alpha_dot[i], self.h_dot[i]
        u, w = self._chord_frame(u1, w1, i)
        af = self.airfoil
        return af['detadx_panel'] * (self.Uinf * np.cos(a) + hd * np.sin(a) + u - ad * af['eta_panel']) \
            - self.Uinf * np.sin(a) - ad * (af['x_panel'] - self.piv) + hd * np.cos(a) - w

    def airfoil_downwash(self, circulation, xw, zw, i):
        """Normal downwash W on the chord at time step i from the given vortices (LUDVM.py:572-595)."""
        g = self.path['airfoil_gamma_points']
        u1, w1 = self.induced_velocity(circulation, xw, zw, g[i, 0, :], g[i, 1, :])
        return self._downwash_from(u1, w1, i)

    # ------------------------------------------------------------------------------------------
    # time loop
    # ------------------------------------------------------------------------------------------
    def _setup_projection(self):
        """Trapezoid weights on theta_panel and the cos/sin tables: integral(y dtheta) = y @ wq,
        A_n = Cproj[n] @ (W/U), sum_n A_n sin(n theta_j) = A[1:] @ Ssin."""
        th = self.airfoil['theta_panel']
        d = np.diff(th)
        wq = np.zeros_like(th)
        wq[:-1] += d / 2
        wq[1:] += d / 2
        n = np.arange(self.Ncoeffs)[:, None]
        cproj = 2 / np.pi * np.cos(n * th[None, :]) * wq
        cproj[0] = -1 / np.pi * wq
        self._wq, self._cproj = wq, cproj
        self._cm1 = (np.cos(th) - 1) * wq
        self._ssin = np.sin(n[1:] * th[None, :])

    def _record_row(self, i):
        return self.history == 'full' or i in self.snapshot_steps or i == self.nt - 1

    def time_loop(self, print_dt=50, BCcheck=False, _resume=None):
        """Time marching (LUDVM.py:597-1171): per step place the new TEV, solve Gamma_TEV (and
        Gamma_LEV when |A0| reaches LESPcrit), rebuild the bound vorticity, integrate the loads and
        convect the wake.  `BCcheck` is accepted for signature compatibility; the reference's check
        (:1144-1161) raises a shape error and has no effect on the results.  `_resume` is the state a
        checkpoint holds (see `resume`).

        The loop itself: every step is MARCHED on the device where the engine offers it (`_march_call`: stretches of steps
        in one ludvm_march_run); the per-step path (`_host_step`: one device round trip, solve on the host) serves
        march=False and engines without the march."""
        S = self._loop_begin()
        if _resume is not None:
            self._loop_restore(S, _resume)
        self._free_slot = S.fslot
        S.fsl = slice(0, S.nf) if S.fslot is None else S.fslot
        self._loop_prepare_engine(S)
        eng, nt = self.engine, self.nt
        try:
            if self._shard is not None:
                # the roll-up's unordered pairs in tile blocks over the ranks, one integer all-reduce per step (attached
                # inside the try: whatever goes wrong from here on, the engine is left unsharded)
                self._shard.attach(eng, S.nf + 2 * (nt - 1) + self.Npoints - 1 + 2)
            i = S.first_step
            while i < nt:
                if S.can_march:
                    j, rec_i = self._march_extent(S, i)
                    if j > i:
                        self._march_call(S, i, j, rec_i, print_dt)
                        i = j
                        continue
                self._host_step(S, i, print_dt)
                i += 1
        finally:
            if self._shard is not None:
                self._shard.detach(eng)
        return None

    # ---- the state a time loop carries, and its set-up --------------------------------------------------------------------
    class _Loop:
        """Scalars carried from step to step (counters, Kelvin sums, the newest shed vortices, the slot maps of the resident
        wake) and the constant tables of a run."""
        __slots__ = (
            # constants of the run
            'nf', 'x_gamma', 'detadx', 'gpts', 'foil', 'one_plus_cos_over_sin', 'half_c_sin_dth', 'wx', 'sum_free', 'first_step',
            'fslot', 'fsl', 'sb', 'prec_code', 'can_march', 'dense_march', 'march_chunk',
            # carried from step to step
            'itev', 'ilev', 'lesp_crit', 'sum_tev', 'sum_lev', 'last_tev', 'last_lev', 'LEV_shed', 'tev_slot', 'lev_slot', 'have_next')

    def _loop_begin(self):
        """Result arrays (LUDVM.py:615-641), projection tables, the device wake with the free vortices in it."""
        pi, U, c = np.pi, self.Uinf, self.chord
        eng = self.engine
        nt, nv, nf, npan = self.nt, self.nt - 1, self.n_freevort, self.Npoints - 1
        S = LUDVM._Loop()
        S.nf = nf
        th, thp = self.airfoil['theta'], self.airfoil['theta_panel']
        S.x_gamma = self.airfoil['x_panel']
        S.detadx = self.airfoil['detadx_panel']
        S.gpts = self.path['airfoil_gamma_points']
        S.foil = self.path['airfoil']
        self._setup_projection()
        dth = th[1:] - th[:-1]
        S.one_plus_cos_over_sin = (1 + np.cos(thp)) / np.sin(thp)
        S.half_c_sin_dth = c / 2 * np.sin(thp) * dth
        S.wx = np.zeros(npan)                       # trapezoid weights on x_gamma (loads, :1071, :1090)
        dxg = np.diff(S.x_gamma)
        S.wx[:-1] += dxg / 2
        S.wx[1:] += dxg / 2

        full = self.history == 'full'
        P = self.path
        if full:
            P['TEV'] = np.zeros([nt, 2, nv])
            P['LEV'] = np.zeros([nt, 2, nv])
            P['FREE'] = np.zeros([nt, 2, nf])
        else:
            P['TEV'], P['LEV'], P['FREE'] = SparseHistory(nt), SparseHistory(nt), SparseHistory(nt)
        free0 = np.array(self.xy_freevort, dtype=float).reshape(2, nf)
        if full:
            P['FREE'][0] = free0
        else:
            P['TEV'].store(0, np.zeros([2, 0]))
            P['LEV'].store(0, np.zeros([2, 0]))
            P['FREE'].store(0, free0.copy())
        C = self.circulation = {'TEV': np.zeros(nv), 'LEV': np.zeros(nv), 'FREE': self.circulation_freevort,
                                'bound': np.zeros(nv), 'airfoil': np.zeros([nv, npan]),
                                'gamma_airfoil': np.zeros([nv, npan]), 'Gamma_airfoil': np.zeros([nv, npan])}
        self.BC = np.zeros([nv, self.Npoints])
        self.dp = np.zeros([nt, npan])
        self.Fn, self.Fs, self.L, self.D, self.T, self.M = (np.zeros(nt) for _ in range(6))
        self.fourier = np.zeros([nt, 2, self.Ncoeffs])
        self.LESP, self.LESP_prev = np.zeros(nt), np.zeros(nt)
        A0, A1 = np.sin(self.alpha_m), 0
        self.fourier[0, 0, :2] = A0, A1
        g_free = np.asarray(self.circulation_freevort, dtype=float)
        S.sum_free = np.sum(C['FREE'])
        C['IC'] = S.sum_free + U * c * pi * (A0 + A1 / 2)

        # device wake, in shedding order: FREE first, then each step's TEV (and LEV when shed)
        eng.wake_clear()
        eng.wake_reserve(nf + 2 * nv + npan + 2)
        # A cloud of free vortices (generate_flowfield_turbulence, LUDVM.py:98-130) arrives in no spatial order; the fp32
        # roll-up keeps its accuracy tier on compact 256-vortex blocks (a shed wake's stored order).  The engine names the
        # order it wants (Morton; the identity for a compact or small set): the cloud is STORED in it, `fslot[j]` is the
        # wake slot of the caller's free vortex j, and every path['FREE'] row is returned in the caller's order.
        fslot = None
        if self.precision == 'f32' and nf >= 2048 and hasattr(eng, 'spatial_order'):
            order, reordered, extent = eng.spatial_order(free0[0], free0[1], with_extent=True)
            if extent > (150.0 if reordered else 300.0) * self.v_core > 0.0:
                # too sparse for its core: no order makes 128-vortex classes compact enough for fp32 offsets to resolve
                # v_core (ludvm_hip.h, ludvm_spatial_order) -- the roll-up takes hi+lo positions instead
                import warnings
                warnings.warn(f"LUDVM: the free-vortex cloud is too sparse for v_core = {self.v_core:g} to keep 1e-5 of max|u| "
                              f"in fp32 on local origins (mean 128-vortex class extent {extent:.3g} > {150 if reordered else 300} v_core); "
                              "the wake-on-wake sums of this run use hi+lo positions (precision='f32x2')",
                              RuntimeWarning, stacklevel=4)
                self.precision, reordered = 'f32x2', False
            if reordered:
                fslot = np.empty(nf, dtype=np.int64)
                fslot[order] = np.arange(nf)
                eng.wake_append(free0[0][order], free0[1][order], g_free[order])
        if fslot is None:
            eng.wake_append(free0[0], free0[1], g_free)
        S.fslot = fslot
        S.tev_slot = np.zeros(nv, dtype=np.int64)
        S.lev_slot = np.zeros(nv, dtype=np.int64)
        S.sum_tev = S.sum_lev = 0.0                 # running Kelvin sums (:758-760)
        S.last_tev = S.last_lev = None              # newest shed vortices after their convection
        S.itev = S.ilev = 0
        S.lesp_crit = self.LESPcrit
        S.LEV_shed = -1 * np.ones(nt)
        S.first_step = 1
        return S

    def _loop_restore(self, S, R):
        """The loop's state as a checkpoint holds it (`_write_checkpoint_file`)."""
        C, P, eng, nf = self.circulation, self.path, self.engine, S.nf
        S.first_step, S.itev, S.ilev = int(R['next_step']), int(R['itev']), int(R['ilev'])
        S.lesp_crit, S.sum_tev, S.sum_lev = float(R['lesp_crit']), float(R['sum_tev']), float(R['sum_lev'])
        S.last_tev = None if np.isnan(R['last_tev']).any() else R['last_tev'].copy()
        S.last_lev = None if np.isnan(R['last_lev']).any() else R['last_lev'].copy()
        S.LEV_shed, S.tev_slot, S.lev_slot = R['LEV_shed'].copy(), R['tev_slot'].copy(), R['lev_slot'].copy()
        for key in ('TEV', 'LEV', 'bound', 'airfoil', 'gamma_airfoil', 'Gamma_airfoil'):
            C[key][...] = R['circ_' + key]
        for name in ('Fn', 'Fs', 'L', 'D', 'T', 'M', 'fourier', 'LESP', 'LESP_prev'):
            getattr(self, name)[...] = R[name]
        for key in ('TEV', 'LEV', 'FREE'):
            if self.history == 'full':
                P[key][:S.first_step] = R['path_' + key]
            else:
                for srow in R['rows_steps']:
                    P[key].store(int(srow), R[f'row_{key}_{int(srow)}'])
        eng.wake_clear()
        eng.wake_append(R['wake_x'], R['wake_z'], R['wake_g'])        # (in the stored order, free vortices included)
        S.fslot = R['free_slot'].astype(np.int64) if ('free_slot' in R.files and R['free_slot'].size == nf) else None
        self.ilev, self.itev, self.LEV_shed = int(R['self_ilev']), int(R['self_itev']), S.LEV_shed

    def _loop_prepare_engine(self, S):
        """Host buffers of the per-step path; tables and kinematics of the device-resident march."""
        eng, C = self.engine, self.circulation
        npan = self.Npoints - 1
        # preallocated host buffers for the two device calls of a step (engines that offer them)
        S.sb = eng.step_buffers(npan) if hasattr(eng, 'step_buffers') else None
        S.prec_code = {'f32': 0, 'f32x2': 1, 'f64': 2}[self.precision]
        S.have_next = False     # sb already holds step i's placement and chord sums (from the previous wake_step)

        # Device-resident march (ludvm_march_setup / ludvm_march_run): stretches of steps run without a host round trip
        # per step; the per-step path serves march=False and engines without the march.
        S.can_march = (self.march and self.method in ('Faure', 'Ramesh') and hasattr(eng, 'march_run') and npan <= 256
                       and 4 <= self.Ncoeffs <= 64)
        if S.can_march:
            tables = np.concatenate([S.detadx, self.airfoil['eta_panel'], S.x_gamma, self._cm1, self._wq, S.one_plus_cos_over_sin,
                                     S.half_c_sin_dth, S.wx, self._cproj.ravel(), self._ssin.ravel()])
            kin = np.concatenate([self.alpha[:, None], self.alpha_dot[:, None], self.h_dot[:, None], S.foil[:, :, -1],
                                  S.foil[:, :, 0], S.gpts[:, 0, :], S.gpts[:, 1, :]], axis=1)
            eng.march_setup(npan, self.Ncoeffs, [self.Uinf, self.chord, self.rho, self.dt, self.piv, self.v_core, C['IC'], S.sum_free,
                                                  float(self.method == 'Ramesh'), self.maxerror, self.maxiter, self.epsilon], tables, kin)
        # with the dense history every step's row is recorded: the march then keeps a snapshot of the wake per step on
        # the device (shorter calls, the snapshots are [steps, 2, wake size])
        S.dense_march = S.can_march and self.history == 'full'
        S.march_chunk = int(getattr(self, '_march_chunk', 512 if S.dense_march else 32768))   # steps per ludvm_march_run call

    def _progress(self, q, print_dt):
        if (q == 1 or q == self.nt - 1 or q / print_dt == int(q / print_dt)) and self.verbose == True:  # noqa: E712
            print('Step {} out of {}. Elapsed time {}'.format(q, self.nt - 1, timeit.default_timer() - self.start_time))

    def _next_placement(self, S, i):
        """New TEV: the first one half a step behind the initial trailing edge, then 1/3 of the way from the trailing edge to
        the previous TEV (:672-681).  Candidate LEV: only geometry and the previous LEV enter (:788-800), so it is known
        before the solve."""
        foil = S.foil
        te, le = foil[i, :, -1], foil[i, :, 0]
        tev_xy = foil[0, :, -1] + np.array([0.5 * self.Uinf * self.dt, 0.0]) if S.itev == 0 else te + (S.last_tev - te) / 3
        lev_xy = le + (S.last_lev - le) / 3 if (S.ilev > 0 and S.LEV_shed[i - 1] != -1) else le.copy()
        return tev_xy, lev_xy

    # ---- marched stretches --------------------------------------------------------------------------------------------------
    def _march_extent(self, S, i):
        """-> (j, rec_i): the stretch [i, j) the next ludvm_march_run call covers.  Every step is marched, whatever the
        history mode: a step whose row is recorded (sparse history) is a call of its own that also returns the wake's
        positions after it -- so a run's bits do not depend on which rows it keeps (rounds 1-3 took recorded steps through
        the per-step path, whose launches are sized from the exact wake size instead of the march's anchor bound: fp32
        rounding differed from the first recorded step on)."""
        nt = self.nt
        rec_i = (not S.dense_march) and self._record_row(i)
        j = i
        if rec_i:
            # a run of consecutive recorded steps is one call (its snapshots are [steps, 2, wake size] doubles:
            # at most 256 MB of them)
            n_now = S.nf + S.itev + S.ilev
            most = max(1, min(512, int(256e6 / (16.0 * (n_now + 1024)))))
            while j < nt and self._record_row(j) and j - i < most:
                j += 1
                if self.checkpoint_every and (j - 1) % self.checkpoint_every == 0:
                    break
        while (not rec_i) and j < nt and (S.dense_march or not self._record_row(j)) and j - i < S.march_chunk:
            j += 1
            if self.checkpoint_every and (j - 1) % self.checkpoint_every == 0:
                break
        return j, rec_i

    def _march_call(self, S, i, j, rec_i, print_dt):
        sb = S.sb
        if S.have_next:
            place = [sb.unit[0, 0], sb.unit[0, 1], sb.unit[1, 0], sb.unit[1, 1]]
        else:
            tev_xy, lev_xy = self._next_placement(S, i)
            place = [tev_xy[0], lev_xy[0], tev_xy[1], lev_xy[1]]
        self._march_stretch(S, i, j, place, record=S.dense_march or rec_i)
        S.have_next = False
        if self.verbose == True:  # noqa: E712
            for q in range(i, j):
                self._progress(q, print_dt)
        if self.checkpoint_every and (j - 1) % self.checkpoint_every == 0 and j - 1 < self.nt - 1:
            self._write_checkpoint(S, j)

    def _march_stretch(self, S, i, j, place, record=False):
        """Time steps [i, j) as one device-resident march (Engine.march_run) and everything the per-step path
        would have stored for them (LUDVM.py:765-1090): circulations, Fourier rows, LESP, loads, slot maps -- and,
        with `record` (dense history), the path[...] rows of every step (:1108-1127).
        `place` = [tev_x, lev_x, tev_z, lev_z] of step i.  The loop state S and the result arrays are updated in place."""
        C, nc, npan, cnt = self.circulation, self.Ncoeffs, self.Npoints - 1, j - i
        nf, itev, ilev, LEV_shed, tev_slot, lev_slot = S.nf, S.itev, S.ilev, S.LEV_shed, S.tev_slot, S.lev_slot
        H = self.engine.MARCH_ROW_HEAD
        n_wake = nf + itev + ilev
        st = np.zeros(16 + nc)
        st[:11] = [n_wake, itev, ilev, float(LEV_shed[i - 1] != -1), S.lesp_crit, S.sum_tev, S.sum_lev] + list(place)
        # wake sizes after the four anchor steps before this call (ludvm_march_run, `anchors`): the launch geometry of
        # every step then follows from the step number and the run itself, not from where the stretches begin
        shed_before = np.cumsum(LEV_shed[:i] != -1)            # LEVs shed in steps 0 .. q (step 0 sheds none)
        anchors = [nf + a + int(shed_before[a]) for a in self.engine.march_anchor_steps(i)]      # (all of them < i)
        st[16:] = self.fourier[i - 1, 0, :]
        if record:
            R, hist = self.engine.march_run(i, cnt, S.prec_code, st, hist_nmax=n_wake + 2 * cnt, anchors=anchors)
        else:
            R = self.engine.march_run(i, cnt, S.prec_code, st, anchors=anchors)
        steps = np.arange(i, j)
        tix = itev + np.arange(cnt)
        shed_v = R[:, 2] != 0
        n_shed = int(shed_v.sum())
        slot = R[:, 9].astype(np.int64)
        if int(st[1]) != itev + cnt or int(st[0]) != n_wake + cnt + n_shed:
            raise RuntimeError("device march returned an inconsistent state")
        C['TEV'][tix] = R[:, 0]
        C['bound'][tix] = R[:, 3]
        self.LESP_prev[tix], self.LESP[tix] = R[:, 4], R[:, 5]
        self.Fn[steps], self.Fs[steps], self.M[steps] = R[:, 6], R[:, 7], R[:, 8]
        ca, sa = np.cos(self.alpha[steps]), np.sin(self.alpha[steps])
        self.L[steps] = self.Fn[steps] * ca + self.Fs[steps] * sa             # :1077-1081
        self.D[steps] = self.Fn[steps] * sa - self.Fs[steps] * ca
        self.T[steps] = -self.D[steps]
        tev_slot[tix] = slot
        levs_before = ilev + np.cumsum(shed_v) - shed_v          # LEVs shed before each step
        lix = levs_before[shed_v]
        C['LEV'][lix] = R[shed_v, 1]
        lev_slot[lix] = slot[shed_v] + 1
        LEV_shed[steps[shed_v]] = lix
        self.fourier[steps, 0, :] = R[:, H:H + nc]
        self.fourier[steps, 1, :] = R[:, H + nc:H + 2 * nc]
        C['gamma_airfoil'][tix] = R[:, H + 2 * nc:H + 2 * nc + npan]
        C['airfoil'][tix] = R[:, H + 2 * nc + npan:]
        C['Gamma_airfoil'][tix] = np.cumsum(C['airfoil'][tix], axis=1)
        if record:
            self._store_marched_rows(S, i, cnt, R, hist, shed_v, levs_before)
        last_shed = bool(shed_v[-1])
        # as the per-step path leaves them: the counters before the last step's increment
        self.itev, self.ilev, self.LEV_shed = itev + cnt - 1, ilev + n_shed - int(last_shed), LEV_shed
        if last_shed:       # newest TEV at size - 2, newest LEV at size - 1
            S.last_tev, S.last_lev = np.array([st[12], st[14]]), np.array([st[13], st[15]])
        else:
            S.last_tev = np.array([st[13], st[15]])
        S.itev, S.ilev = itev + cnt, ilev + n_shed
        S.lesp_crit, S.sum_tev, S.sum_lev = float(st[4]), float(st[5]), float(st[6])

    def _store_marched_rows(self, S, i, cnt, R, hist, shed_v, levs_before):
        """History rows from the march's per-step snapshots (wake order -> TEV / LEV / FREE slots); on a step without LEV
        shedding the reference's zero-strength LEV slot lands at dt * (velocity at the origin)."""
        P, tev_slot, lev_slot, fsl = self.path, S.tev_slot, S.lev_slot, S.fsl
        for r in range(cnt):
            q, it, il = i + r, S.itev + r, int(levs_before[r])
            xs, zs = hist[r, 0], hist[r, 1]
            ts = tev_slot[:it + 1]
            ls = lev_slot[:il + 1] if shed_v[r] else lev_slot[:il]
            if self.history == 'full':
                P['TEV'][q, 0, :it + 1], P['TEV'][q, 1, :it + 1] = xs[ts], zs[ts]
                P['LEV'][q, 0, :len(ls)], P['LEV'][q, 1, :len(ls)] = xs[ls], zs[ls]
                if not shed_v[r]:
                    P['LEV'][q, :, il] = self.dt * R[r, 10:12]
                P['FREE'][q, 0, :], P['FREE'][q, 1, :] = xs[fsl], zs[fsl]
            elif self._record_row(q):
                # sparse history: the rows the per-step path would have stored (LEV row with the zero-strength slot)
                row_l = np.stack([xs[ls], zs[ls]])
                if not shed_v[r]:
                    row_l = np.concatenate([row_l, (self.dt * R[r, 10:12])[:, None]], axis=1)
                P['TEV'].store(q, np.stack([xs[ts], zs[ts]]))
                P['LEV'].store(q, row_l)
                P['FREE'].store(q, np.stack([xs[fsl], zs[fsl]]))

    # ---- the per-step path: one device round trip, solve on the host --------------------------------------------------------
    def _host_step(self, S, i, print_dt):
        self._progress(i, print_dt)
        n_wake = S.nf + S.itev + S.ilev
        tev_xy, lev_xy, sums = self._chord_sums(S, i, n_wake)
        g_tev, g_lev, shed, A = self._solve_circulations(S, i, sums)
        dGamma = self._bound_vorticity_and_loads(S, i, sums, g_tev, g_lev, shed, A)
        self._roll_up(S, i, n_wake, tev_xy, lev_xy, g_tev, g_lev, shed, dGamma)
        S.sum_tev += g_tev
        self.ilev, self.itev, self.LEV_shed = S.ilev, S.itev, S.LEV_shed
        if shed:
            S.sum_lev += g_lev
            S.ilev += 1
        S.itev += 1
        if self.checkpoint_every and i % self.checkpoint_every == 0 and i < self.nt - 1:
            self._write_checkpoint(S, i + 1)

    def _chord_sums(self, S, i, n_wake):
        """Existing wake -> chord (T1), unit new TEV -> chord (T2), unit candidate LEV -> chord (T3): one round trip
        (:743-754, :924-934).  -> (tev_xy, lev_xy, (u1, w1, ut1, wt1, ul1, wl1)) in the global frame."""
        eng, sb = self.engine, S.sb
        xg, zg = S.gpts[i, 0, :], S.gpts[i, 1, :]
        if S.have_next:
            # placed on the device right after the previous roll-up, sums already here
            tev_xy = np.array([sb.unit[0, 0], sb.unit[1, 0]])
            lev_xy = np.array([sb.unit[0, 1], sb.unit[1, 1]])
            return tev_xy, lev_xy, (sb.u, sb.w, sb.uu[0], sb.wu[0], sb.uu[1], sb.wu[1])
        tev_xy, lev_xy = self._next_placement(S, i)
        if sb is not None:
            sb.unit[0, 0], sb.unit[0, 1], sb.unit[1, 0], sb.unit[1, 1] = tev_xy[0], lev_xy[0], tev_xy[1], lev_xy[1]
            eng.wake_chord_sums_into(sb, n_wake, xg, zg, float(self.v_core))
            return tev_xy, lev_xy, (sb.u, sb.w, sb.uu[0], sb.wu[0], sb.uu[1], sb.wu[1])
        u1, w1, uu, wu = eng.wake_chord_sums(0, n_wake, xg, zg, [tev_xy[0], lev_xy[0]], [tev_xy[1], lev_xy[1]], self.v_core)
        return tev_xy, lev_xy, (u1, w1, uu[0], wu[0], uu[1], wu[1])

    def _solve_circulations(self, S, i, sums):
        """Gamma_TEV from Kelvin's condition (:758-760, or Ramesh's Newton :683-739), the Fourier projection (:765-773), the
        LESP test (:781) and, when it trips, Gamma_TEV / Gamma_LEV together (:944-954, :807-914).
        -> (g_tev, g_lev, shed, A)."""
        pi, U, c, dt = np.pi, self.Uinf, self.chord, self.dt
        C, itev, ilev = self.circulation, S.itev, S.ilev
        wq, cproj, cm1, detadx = self._wq, self._cproj, self._cm1, S.detadx
        u1, w1, ut1, wt1, ul1, wl1 = sums
        T1 = self._downwash_from(u1, w1, i)
        ut, un = self._chord_frame(ut1, wt1, i)
        T2 = detadx * ut - un
        I1, I2 = T1 @ cm1, T2 @ cm1
        kelvin = S.sum_tev + S.sum_lev + S.sum_free - C['IC']
        if self.method == 'Ramesh':
            g_tev = self._newton_tev(T1, T2, kelvin)                         # :683-739
        elif self.method == 'Faure':
            g_tev = -(I1 + kelvin) / (1 + I2)                                # :758-760
        else:
            raise ValueError("method must be 'Faure' or 'Ramesh'")
        g_lev = 0.0
        W = T1 + g_tev * T2
        A = cproj @ (W / U)
        C['bound'][itev] = I1 + g_tev * I2 if self.method == 'Faure' else U * c * pi * (A[0] + A[1] / 2)
        self.fourier[i, 0, :] = A
        self.fourier[i, 1, :] = (A - self.fourier[i - 1, 0, :]) / dt          # :772-773
        self.LESP_prev[itev] = A[0]

        shed = abs(A[0]) >= abs(S.lesp_crit)                                  # :781
        if shed:
            S.LEV_shed[i] = ilev
            S.lesp_crit = -abs(S.lesp_crit) if A[0] < 0 else abs(S.lesp_crit)   # :802-805
            ult, uln = self._chord_frame(ul1, wl1, i)
            T3 = detadx * ult - uln
            I3 = T3 @ cm1
            J1, J2, J3 = (-1 / pi * (T @ wq) for T in (T1, T2, T3))
            if self.method == 'Ramesh':
                g_tev, g_lev = self._newton_tev_lev(T1, T2, T3, kelvin, S.lesp_crit, g_tev)   # :807-914
            else:
                g_tev, g_lev = np.linalg.solve(np.array([[1 + I2, 1 + I3], [J2, J3]]),
                                               np.array([-(I1 + kelvin), S.lesp_crit - J1]))  # :944-954
            W = T1 + g_tev * T2 + g_lev * T3
            A = cproj @ (W / U)
            if self.method == 'Faure':
                C['bound'][itev] = I1 + g_tev * I2 + g_lev * I3
                A[0] = J1 + g_tev * J2 + g_lev * J3                           # :959
            else:
                C['bound'][itev] = U * c * pi * (A[0] + A[1] / 2)
            self.fourier[i, 0, :] = A       # derivatives keep their pre-LEV values (:963-966)
            C['LEV'][ilev] = g_lev
        C['TEV'][itev] = g_tev
        self.LESP[itev] = A[0]
        return g_tev, g_lev, shed, A

    def _bound_vorticity_and_loads(self, S, i, sums, g_tev, g_lev, shed, A):
        """Bound vorticity per panel (:987-1010) and the loads (:1035-1090).  The tangential velocity on the chord from the
        full wake (incl. the new TEV/LEV with their solved circulations) follows by linearity from u1 and the unit
        influences already evaluated: no further pair sum.  -> dGamma, the panel circulations."""
        pi, U, c, rho = np.pi, self.Uinf, self.chord, self.rho
        C, itev = self.circulation, S.itev
        u1, w1, ut1, wt1, ul1, wl1 = sums
        A0, A1, A2 = A[0], A[1], A[2]
        A0d, A1d, A2d, A3d = self.fourier[i, 1, :4]
        gamma = 2 * U * (A0 * S.one_plus_cos_over_sin + A[1:] @ self._ssin)
        dGamma = gamma * S.half_c_sin_dth
        C['airfoil'][itev], C['gamma_airfoil'][itev] = dGamma, gamma
        C['Gamma_airfoil'][itev] = np.cumsum(dGamma)

        uc1, wc1 = u1 + g_tev * ut1, w1 + g_tev * wt1
        if shed:
            uc1, wc1 = uc1 + g_lev * ul1, wc1 + g_lev * wl1
        u, _ = self._chord_frame(uc1, wc1, i)
        a, hd = self.alpha[i], self.h_dot[i]
        Ueff = U * np.cos(a) + hd * np.sin(a)
        self.Fn[i] = rho * pi * c * U * (Ueff * (A0 + 0.5 * A1) + c * (3 / 4 * A0d + 1 / 4 * A1d + 1 / 8 * A2d)) \
            + rho * ((u * gamma) @ S.wx)
        self.Fs[i] = rho * pi * c * U**2 * A0**2
        self.L[i] = self.Fn[i] * np.cos(a) + self.Fs[i] * np.sin(a)
        self.D[i] = self.Fn[i] * np.sin(a) - self.Fs[i] * np.cos(a)
        self.T[i] = -self.D[i]
        self.M[i] = self.piv * self.Fn[i] - rho * pi * c**2 * U * (
            Ueff * (1 / 4 * A0 + 1 / 4 * A1 - 1 / 8 * A2)
            + c * (7 / 16 * A0d + 3 / 16 * A1d + 1 / 16 * A2d - 1 / 64 * A3d)) \
            - rho * ((u * gamma * S.x_gamma) @ S.wx)
        return dGamma

    def _roll_up(self, S, i, n_wake, tev_xy, lev_xy, g_tev, g_lev, shed, dGamma):
        """Wake roll-up (:1095-1127): shed vortices join the resident wake, then one fused launch (wake + bound vortices ->
        every wake vortex, explicit Euler update on the device); the history row of a recorded step."""
        eng, sb, P, nt = self.engine, S.sb, self.path, self.nt
        itev, ilev, tev_slot, lev_slot = S.itev, S.ilev, S.tev_slot, S.lev_slot
        dt, vc = self.dt, self.v_core
        xg, zg = S.gpts[i, 0, :], S.gpts[i, 1, :]
        record = self._record_row(i)
        new_x, new_z, new_g = [tev_xy[0]], [tev_xy[1]], [g_tev]
        tev_slot[itev] = n_wake
        if shed:
            lev_slot[ilev] = n_wake + 1
            new_x.append(lev_xy[0]); new_z.append(lev_xy[1]); new_g.append(g_lev)
        elif record:
            # the reference also convects LEV slot `ilev` (zero strength, at the origin) on a
            # non-shedding step and stores it in path['LEV'][i]; reproduce that row entry
            new_x.append(0.0); new_z.append(0.0); new_g.append(0.0)
        n_after = n_wake + len(new_x)
        one_trip = (not record) and sb is not None and i < nt - 1 and hasattr(eng, 'wake_step_into')
        if not one_trip:
            eng.wake_append(new_x, new_z, new_g)

        if record:
            S.have_next = False
            eng.wake_advect(dt, xg, zg, dGamma, vc, precision=self.precision)
            xs, zs = eng.wake_read(0, n_after)
            row_t = np.stack([xs[tev_slot[:itev + 1]], zs[tev_slot[:itev + 1]]])
            lslots = lev_slot[:ilev + 1] if shed else np.append(lev_slot[:ilev], n_after - 1)
            row_l = np.stack([xs[lslots], zs[lslots]])
            row_f = np.stack([xs[S.fsl], zs[S.fsl]])
            if self.history == 'full':
                P['TEV'][i, :, :itev + 1] = row_t
                P['LEV'][i, :, :ilev + 1] = row_l
                P['FREE'][i] = row_f
            else:
                P['TEV'].store(i, row_t)
                P['LEV'].store(i, row_l)
                P['FREE'].store(i, row_f)
            S.last_tev = row_t[:, -1].copy()
            if shed:
                S.last_lev = row_l[:, -1].copy()
            elif len(new_x) == 2:
                eng.wake_truncate(n_after - 1)   # drop the phantom LEV slot
            return
        # only the newest TEV / LEV come back: they place the next ones (:680-681, :797-798)
        k = 2 if shed else 1
        S.have_next = False
        dt_f, vc_f = float(dt), float(vc)
        if one_trip:
            # append of the shed vortices + roll-up of this step + placement and chord sums of the
            # next one: one packed upload, one download
            foil, gpts = S.foil, S.gpts
            eng.wake_step_into(sb, np.array(new_x), np.array(new_z), np.array(new_g, dtype=float), dt_f, xg, zg,
                               dGamma, vc_f, S.prec_code, np.ascontiguousarray(foil[i + 1, :, -1]),
                               np.ascontiguousarray(foil[i + 1, :, 0]), shed, k, gpts[i + 1, 0, :],
                               gpts[i + 1, 1, :])
            xs, zs = sb.tail[0], sb.tail[1]
            S.have_next = True
        elif sb is not None:
            eng.wake_advect_tail_into(sb, dt_f, xg, zg, dGamma, vc_f, k, S.prec_code)
            xs, zs = sb.tail[0], sb.tail[1]
        else:
            xs, zs = eng.wake_advect_tail(dt, xg, zg, dGamma, vc, k, precision=self.precision)
        S.last_tev = np.array([xs[0], zs[0]])
        if shed:
            S.last_lev = np.array([xs[1], zs[1]])

    # ------------------------------------------------------------------------------------------
    # checkpoint / resume (not in the reference)
    # ------------------------------------------------------------------------------------------
    def _write_checkpoint(self, S, next_step):
        # one simulation shared by several ranks: every rank holds the same state, rank 0 alone writes it (they would race on
        # the temporary file), and nobody goes on before the file is in place
        sh = self._shard if (self._shard is not None and self._shard.world > 1) else None
        if sh is None:
            return self._write_checkpoint_file(S, next_step)
        # The ranks meet in all_ok() -- the barrier -- and learn there whether the file is in place: if rank 0 could not
        # write it (disk full, bad path) everybody raises, instead of rank 0 raising alone and the others waiting for it in
        # a collective that has no timeout
        err = None
        if sh.rank == 0:
            try:
                self._write_checkpoint_file(S, next_step)
            except Exception as e:          # noqa: BLE001  (re-raised below, on every rank)
                err = e
        if not sh.all_ok(err is None):
            if err is not None:
                raise err
            raise RuntimeError(f"rank 0 could not write the checkpoint {self.checkpoint_path}")

    def _write_checkpoint_file(self, S, next_step):
        import json
        import os
        C, P = self.circulation, self.path
        itev, ilev, lesp_crit, sum_tev, sum_lev = S.itev, S.ilev, S.lesp_crit, S.sum_tev, S.sum_lev
        last_tev, last_lev, LEV_shed, tev_slot, lev_slot = S.last_tev, S.last_lev, S.LEV_shed, S.tev_slot, S.lev_slot
        n = self.engine.wake_size()
        wx, wz, wg = self.engine.wake_read(0, n, gamma=True)
        nan2 = np.full(2, np.nan)
        d = dict(ctor=np.array(json.dumps(self._ctor)), next_step=next_step, itev=itev, ilev=ilev, lesp_crit=lesp_crit,
                 sum_tev=sum_tev, sum_lev=sum_lev, last_tev=nan2 if last_tev is None else last_tev,
                 last_lev=nan2 if last_lev is None else last_lev, LEV_shed=LEV_shed, tev_slot=tev_slot,
                 lev_slot=lev_slot, self_itev=self.itev, self_ilev=self.ilev, wake_x=wx, wake_z=wz, wake_g=wg,
                 circulation_freevort=np.asarray(self.circulation_freevort),
                 free_slot=np.zeros(0, np.int64) if self._free_slot is None else self._free_slot,
                 xy_freevort=np.asarray(self.xy_freevort, dtype=float))
        for key in ('TEV', 'LEV', 'bound', 'airfoil', 'gamma_airfoil', 'Gamma_airfoil'):
            d['circ_' + key] = C[key]
        for name in ('Fn', 'Fs', 'L', 'D', 'T', 'M', 'fourier', 'LESP', 'LESP_prev'):
            d[name] = getattr(self, name)
        if self.history == 'full':
            for key in ('TEV', 'LEV', 'FREE'):
                d['path_' + key] = P[key][:next_step]
        else:
            steps = P['TEV'].steps()
            d['rows_steps'] = np.array(steps, dtype=np.int64)
            for key in ('TEV', 'LEV', 'FREE'):
                for srow in steps:
                    d[f'row_{key}_{srow}'] = P[key][srow]
        tmp = self.checkpoint_path + '.tmp.npz'
        np.savez(tmp, **d)
        os.replace(tmp, self.checkpoint_path)      # a reader never sees a half-written file

    @classmethod
    def resume(cls, path, engine=None, device=0, verbose=True, checkpoint_every=0, checkpoint_path=None, march=True,
               distributed=None, devices=None):
        """Continue a run from a checkpoint written with `checkpoint_every` / `checkpoint_path`: rebuilds
        geometry and kinematics from the stored constructor arguments, uploads the wake and marches
        from the stored step to the end.  `distributed` as in the constructor: every rank of the group resumes from the
        same file; `devices` as in the constructor: one process, a replica per device, each resumed from the file."""
        import json
        if devices is not None:
            from .multi import MultiDeviceLUDVM, normalise_devices
            devs = normalise_devices(devices)
            if len(devs) > 1:
                if engine is not None or distributed is not None:
                    raise ValueError("devices=[...] creates the engines and their communicator itself: do not pass engine= / distributed=")
                return MultiDeviceLUDVM((), {}, devs, builder=lambda r, eng, grp: cls.resume(
                    path, engine=eng, verbose=verbose and r == 0, checkpoint_every=checkpoint_every, checkpoint_path=checkpoint_path,
                    march=march, distributed=grp))
            device = devs[0]
        R = np.load(path, allow_pickle=False)
        kw = json.loads(str(R['ctor']))
        free = {}
        if R['circulation_freevort'].size != 1 or float(np.abs(R['circulation_freevort']).sum()) != 0.0 \
                or float(np.abs(R['xy_freevort']).sum()) != 0.0:
            free = dict(circulation_freevort=R['circulation_freevort'], xy_freevort=R['xy_freevort'])
        sim = cls(**kw, **free, verbose=verbose, engine=engine, device=device, run=False,
                  checkpoint_every=checkpoint_every, checkpoint_path=checkpoint_path, march=march, distributed=distributed)
        sim.time_loop(_resume=R)
        sim.compute_coefficients()
        return sim

    # Newton variants of the reference's 'Ramesh' method.  The downwash is linear in the circulations
    # of the vortices being shed, W = T1 + G_tev T2 + G_lev T3, so the reference's repeated
    # airfoil_downwash evaluations (:692, :707, :820, :837, :852) need no further pair sums here.
    def _kelvin_residual(self, W, kelvin, g_new):
        A0 = self._cproj[0] @ (W / self.Uinf)
        A1 = self._cproj[1] @ (W / self.Uinf)
        return self.Uinf * self.chord * np.pi * (A0 + A1 / 2) + kelvin + g_new, A0

    def _newton_tev(self, T1, T2, kelvin):
        eps = self.epsilon
        f, niter, g = 1.0, 1, -1.0
        while abs(f) > self.maxerror and niter < self.maxiter:
            f, _ = self._kelvin_residual(T1 + g * T2, kelvin, g)
            fd, _ = self._kelvin_residual(T1 + (g + eps) * T2, kelvin, g + eps)
            g = g - f / ((fd - f) / eps)
            niter += 1
        if niter >= self.maxiter:
            print('The solution did not converge during the Newton-Raphson iteration')
        return g

    def _newton_tev_lev(self, T1, T2, T3, kelvin, lesp_crit, guess):
        eps = self.epsilon
        g_tev = g_lev = guess
        f1 = f2 = 0.1
        niter = 1

        def res(gt, gl):
            f, A0 = self._kelvin_residual(T1 + gt * T2 + gl * T3, kelvin, gt + gl)
            return f, lesp_crit - A0

        while (abs(f1) > self.maxerror or abs(f2) > self.maxerror) and niter < self.maxiter:
            f1, f2 = res(g_tev, g_lev)
            f1t, f2t = res(g_tev + eps, g_lev)
            f1l, f2l = res(g_tev, g_lev + eps)
            J = np.array([[(f1l - f1) / eps, (f1t - f1) / eps], [(f2l - f2) / eps, (f2t - f2) / eps]])
            g_lev, g_tev = np.array([g_lev, g_tev]) - np.linalg.solve(J, np.array([f1, f2]))
            niter += 1
        if niter >= self.maxiter:
            print('The solution did not converge when solving the LEV-TEV nonlinear system')
        return g_tev, g_lev

    def compute_coefficients(self):
        """Force and moment coefficients (LUDVM.py:1173-1184)."""
        q = 0.5 * self.rho * self.Uinf**2
        qc = q * self.chord
        self.Cp = self.dp / q
        self.Cn, self.Cs = self.Fn / qc, self.Fs / qc
        self.Cl, self.Cd, self.Ct = self.L / qc, self.D / qc, self.T / qc
        self.Cm = self.M / (qc * self.chord)
        return None

    # ------------------------------------------------------------------------------------------
    # flow field
    # ------------------------------------------------------------------------------------------
    def _flowfield_sources(self, s):
        """Sources the reference gathers for time step s (LUDVM.py:1202-1215), index quirks kept:
        TEV/LEV positions from row s-1 with slots [:s+1] / [:ilev+1] (not-yet-shed slots sit at the
        origin but carry their final circulation), FREE from row s, every LEV dropped when
        LEV_shed[s] == -1; bound vortices of step s-1."""
        C, P = self.circulation, self.path
        g_free = np.asarray(C['FREE'], dtype=float)
        if s == 0:
            f0 = P['FREE'][0]
            return g_free, f0[0], f0[1]
        ilev = int(self.LEV_shed[s])

        def padded(row, n):
            out = np.zeros([2, n])
            m = min(n, row.shape[1])
            out[:, :m] = row[:, :m]
            return out
        tev = padded(P['TEV'][s - 1], s + 1)
        lev = padded(P['LEV'][s - 1], ilev + 1)
        free = P['FREE'][s]
        gp = self.path['airfoil_gamma_points'][s - 1]
        g = np.concatenate([C['TEV'][:s + 1], C['LEV'][:ilev + 1], g_free, C['airfoil'][s - 1, :]])
        xw = np.concatenate([tev[0], lev[0], free[0], gp[0]])
        zw = np.concatenate([tev[1], lev[1], free[1], gp[1]])
        return g, xw, zw

    @staticmethod
    def flowfield_rows_needed(tsteps):
        """History rows `flowfield(tsteps=...)` reads: the reference takes TEV / LEV positions from row s - 1 and FREE
        positions from row s (LUDVM.py:1212-1213).  A run with history='sparse' must have recorded them:
        LUDVM(..., snapshot_steps=LUDVM.flowfield_rows_needed(tsteps))."""
        return sorted({r for s in tsteps for r in ((int(s) - 1, int(s)) if int(s) > 0 else (0,))})

    def flowfield(self, xmin=-10, xmax=0, zmin=-4, zmax=4, dr=0.02, tsteps=[0, 1, 2]):
        """Velocity and vorticity on the uniform mesh arange(xmin, xmax, dr) x arange(zmin, zmax, dr)
        at the requested time steps (LUDVM.py:1186-1298).  The mesh points are generated on the
        device (x-major, as meshgrid(indexing='ij') ravels, :1194-1195); wake and bound vortices go in
        one launch; the vorticity stencil (:1224-1292) runs on the device on the velocity fields where they
        are, and the three fields come back together.  A run in precision 'f64' (what 'auto' picks for every case whose
        dense history the reference itself could hold) evaluates the field in float64 throughout, as the reference does
        (:1206, :1216-1217): u_ff, w_ff and ome_ff then equal the reference's to rounding; 'f32' / 'f32x2' runs evaluate
        it in fp32 on local-origin sources (~1e-6 of max|u|; the stencil amplifies that by 1 / (2 dr) in ome_ff)."""
        x1, z1 = np.arange(xmin, xmax, dr), np.arange(zmin, zmax, dr)
        nx, nz = len(x1), len(z1)
        x, z = np.meshgrid(x1, z1, indexing='ij')
        nsteps = len(tsteps)
        u = np.zeros([nsteps, nx, nz])
        w = np.zeros([nsteps, nx, nz])
        ome = np.zeros([nsteps, nx, nz])
        fused = hasattr(self.engine, 'flowfield_vorticity')
        sh = self._shard if (self._shard is not None and self._shard.world > 1) else None
        ffp = 'f64' if self.precision == 'f64' else 'f32'
        for ii, s in enumerate(tsteps):
            if self.verbose:
                print('Flowfield tstep =', s)
            try:
                g, xw, zw = self._flowfield_sources(int(s))
            except KeyError as e:
                raise KeyError(f"flowfield(tsteps=[{int(s)}]) reads the history rows of steps {int(s) - 1} and {int(s)}, "
                               f"which this run (history='sparse') did not record: construct it with "
                               f"snapshot_steps=LUDVM.flowfield_rows_needed(tsteps) = "
                               f"{self.flowfield_rows_needed(tsteps)}") from e
            if sh is not None:
                # grid rows in contiguous blocks over the ranks, nothing exchanged until the finished rows are gathered
                r0, r1, _ = sh.block(nx)
                if hasattr(self.engine, 'flowfield_rows'):
                    ub, wb, ob = self.engine.flowfield_rows(xmin, zmin, dr, nx, nz, r0, r1 - r0, g, xw, zw, self.v_core,
                                                            precision=ffp)
                else:       # engines without the row entry: the block with one halo row per interior side
                    h0, h1 = max(0, r0 - 1), min(nx, r1 + 1)
                    uh, wh = self.engine.flowfield(x1[h0] if h1 > h0 else xmin, zmin, dr, h1 - h0, nz, g, xw, zw, self.v_core)
                    oh = self.engine.vorticity(uh, wh, dr) if h1 - h0 >= 2 else np.zeros_like(uh)
                    ub, wb, ob = uh[r0 - h0:r1 - h0], wh[r0 - h0:r1 - h0], oh[r0 - h0:r1 - h0]
                fields = sh.gather_blocks(np.stack([ub, wb, ob], axis=1), nx)     # [nx, 3, nz]
                u[ii], w[ii], ome[ii] = fields[:, 0], fields[:, 1], fields[:, 2]
            elif fused:
                u[ii], w[ii], ome[ii] = self.engine.flowfield_vorticity(xmin, zmin, dr, nx, nz, g, xw, zw, self.v_core,
                                                                        precision=ffp)
            else:
                uf, wf = self.engine.flowfield(xmin, zmin, dr, nx, nz, g, xw, zw, self.v_core)
                u[ii], w[ii] = uf, wf
                ome[ii] = self.engine.vorticity(uf, wf, dr)
        self.x_ff, self.z_ff = x, z
        self.u_ff, self.w_ff = u, w
        self.ome_ff = ome
        return None
