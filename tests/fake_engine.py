"""TEST-ONLY stand-in for ludvm_amd.engine.Engine: same method surface, pair sums by the CPU oracle.

Lets the CPU test tier pin the *host logic* of the drop-in class (vortex placement, Gamma solve,
Fourier projection, loads, history bookkeeping, flow-field gathers) against the golden vectors
without a GPU.  It lives under tests/ and is never importable from the product package.
"""
import numpy as np

from oracle import ludvm_oracle as O


class FakeEngine:
    def __init__(self):
        self.x = np.zeros(0)
        self.z = np.zeros(0)
        self.g = np.zeros(0)
        self.calls = {"induce": 0, "points": 0, "advect": 0, "chord": 0}

    # stateless
    def induce(self, circulation, xw, zw, xp, zp, v_core, precision="f32"):
        self.calls["induce"] += 1
        with np.errstate(all="ignore"):
            return O.induced_velocity(np.asarray(circulation, float), xw, zw, xp, zp, v_core)

    # resident wake
    def wake_clear(self):
        self.x, self.z, self.g = np.zeros(0), np.zeros(0), np.zeros(0)

    def wake_reserve(self, capacity):
        pass

    def wake_size(self):
        return len(self.x)

    def wake_truncate(self, n):
        assert n <= len(self.x)
        self.x, self.z, self.g = self.x[:n], self.z[:n], self.g[:n]

    def wake_append(self, x, z, gamma):
        self.x = np.concatenate([self.x, np.asarray(x, float).reshape(-1)])
        self.z = np.concatenate([self.z, np.asarray(z, float).reshape(-1)])
        self.g = np.concatenate([self.g, np.asarray(gamma, float).reshape(-1)])

    def wake_write(self, first, x=None, z=None, gamma=None):
        for dst, src in ((self.x, x), (self.z, z), (self.g, gamma)):
            if src is not None:
                src = np.asarray(src, float).reshape(-1)
                dst[first:first + len(src)] = src

    def wake_read(self, first, count, gamma=False):
        s = slice(first, first + count)
        return (self.x[s].copy(), self.z[s].copy(), self.g[s].copy()) if gamma else (self.x[s].copy(), self.z[s].copy())

    def wake_induce_on_points(self, src_first, src_count, xp, zp, v_core):
        self.calls["points"] += 1
        s = slice(src_first, src_first + src_count)
        return O.induced_velocity(self.g[s], self.x[s], self.z[s], xp, zp, v_core)

    def wake_chord_sums(self, src_first, src_count, xp, zp, unit_x, unit_z, v_core):
        self.calls["chord"] += 1
        s = slice(src_first, src_first + src_count)
        u, w = O.induced_velocity(self.g[s], self.x[s], self.z[s], xp, zp, v_core)
        uu = np.empty([len(unit_x), len(xp)])
        wu = np.empty_like(uu)
        for k in range(len(unit_x)):
            uu[k], wu[k] = O.induced_velocity(np.array([1]), np.array([unit_x[k]]), np.array([unit_z[k]]), xp, zp, v_core)
        return u, w, uu, wu

    def wake_advect_tail(self, dt, foil_x, foil_z, foil_dgamma, v_core, tail_count, precision="f32"):
        self.wake_advect(dt, foil_x, foil_z, foil_dgamma, v_core, precision)
        return self.x[-tail_count:].copy(), self.z[-tail_count:].copy()

    # preallocated-buffer variants used by the time loop's fast path
    def step_buffers(self, npoints):
        b = type("StepBuffers", (), {})()
        b.n = npoints
        b.unit = np.zeros([2, 2])
        b.u, b.w = np.empty(npoints), np.empty(npoints)
        b.uu, b.wu = np.empty([2, npoints]), np.empty([2, npoints])
        b.tail = np.empty([2, 2])
        return b

    def wake_chord_sums_into(self, b, src_count, xp, zp, v_core):
        b.u[:], b.w[:], uu, wu = self.wake_chord_sums(0, src_count, xp, zp, b.unit[0], b.unit[1], v_core)
        b.uu[:], b.wu[:] = uu, wu

    def wake_advect_tail_into(self, b, dt, foil_x, foil_z, foil_dgamma, v_core, tail_count, precision):
        tx, tz = self.wake_advect_tail(dt, foil_x, foil_z, foil_dgamma, v_core, tail_count)
        b.tail[0, :tail_count], b.tail[1, :tail_count] = tx, tz

    def wake_step_into(self, b, new_x, new_z, new_gamma, dt, foil_x, foil_z, foil_dgamma, v_core, precision, te, le,
                       lev_from_prev, tail_count, xp_next, zp_next):
        self.calls["step"] = self.calls.get("step", 0) + 1
        self.wake_append(new_x, new_z, new_gamma)
        self.wake_advect(dt, foil_x, foil_z, foil_dgamma, v_core)
        n = len(self.x)
        b.tail[0, :tail_count], b.tail[1, :tail_count] = self.x[n - tail_count:], self.z[n - tail_count:]
        it = n - tail_count
        b.unit[0, 0] = te[0] + (self.x[it] - te[0]) / 3
        b.unit[1, 0] = te[1] + (self.z[it] - te[1]) / 3
        if lev_from_prev and tail_count == 2:
            b.unit[0, 1] = le[0] + (self.x[n - 1] - le[0]) / 3
            b.unit[1, 1] = le[1] + (self.z[n - 1] - le[1]) / 3
        else:
            b.unit[0, 1], b.unit[1, 1] = le[0], le[1]
        self.wake_chord_sums_into(b, n, xp_next, zp_next, v_core)

    def wake_advect(self, dt, foil_x, foil_z, foil_dgamma, v_core, precision="f32", return_velocity=False):
        self.calls["advect"] += 1
        g = np.concatenate([self.g, np.asarray(foil_dgamma, float)])
        xs = np.concatenate([self.x, np.asarray(foil_x, float)])
        zs = np.concatenate([self.z, np.asarray(foil_z, float)])
        u, w = O.induced_velocity(g, xs, zs, self.x, self.z, v_core)
        self.x = self.x + dt * u
        self.z = self.z + dt * w
        return (u, w) if return_velocity else None

    # flow field
    def flowfield(self, xmin, zmin, dr, nx, nz, circulation, xw, zw, v_core):
        xg = xmin + np.arange(nx) * dr
        zg = zmin + np.arange(nz) * dr
        X, Z = np.meshgrid(xg, zg, indexing="ij")
        u, w = O.induced_velocity(circulation, xw, zw, X.ravel(), Z.ravel(), v_core, rows_per_chunk=4096)
        return u.reshape(nx, nz), w.reshape(nx, nz)

    def vorticity(self, u, w, dr):
        nx, nz = u.shape
        X, Z = np.meshgrid(np.arange(nx) * dr, np.arange(nz) * dr, indexing="ij")
        return O.vorticity(u[None], w[None], X, Z)[0]
