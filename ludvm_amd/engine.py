"""Thin Python handle on one libludvm_hip context (one GPU, one stream).

Normalises what Python callers pass (any array-like, any dtype, any stride -- the reference's
induced_velocity takes non-contiguous views and integer circulations, LUDVM.py:549-570, :751) into
the contiguous float64 / float32 buffers the C ABI takes, and turns status codes into exceptions.
"""
import ctypes
from ctypes import POINTER, byref, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

import numpy as np

from . import _ffi
from ._ffi import PREC_F32, PREC_F32X2, PREC_F64, LudvmHipError

PRECISIONS = {"f32": PREC_F32, "f32x2": PREC_F32X2, "f64": PREC_F64}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def _pd(a):
    return a.ctypes.data_as(POINTER(c_double)) if a is not None else None


def _pf(a):
    return a.ctypes.data_as(POINTER(c_float)) if a is not None else None


def _prec(p):
    if isinstance(p, str):
        return PRECISIONS[p]
    return int(p)


class Engine:
    """One HIP context.  Not thread-safe (one host thread at a time), like the reference."""

    def __init__(self, device=0, lib_path=None):
        self._lib = _ffi.load(lib_path)
        self._ctx = c_void_p()
        rc = self._lib.ludvm_create(int(device), byref(self._ctx))
        if rc != _ffi.OK:
            self._ctx = c_void_p()
            raise LudvmHipError(rc, {_ffi.E_NODEVICE: "no usable gfx950 device (MI355X required)",
                                     _ffi.E_ARG: "bad device ordinal"}.get(rc, "ludvm_create failed"))
        self.device = int(device)

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _ffi.OK:
            msg = self._lib.ludvm_last_error(self._ctx)
            err = LudvmHipError(rc, msg.decode() if msg else "")
            cause = getattr(self, "_hook_error", None)      # an exception raised inside the all-reduce hook (set_shard)
            if cause is not None:
                self._hook_error = None
                raise err from cause
            raise err

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.ludvm_destroy(self._ctx)
            self._ctx = c_void_p()
            self._hook_c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self):
        cu, khz, mem = c_int(), c_int(), c_longlong()
        name = ctypes.create_string_buffer(256)
        self._check(self._lib.ludvm_device_info(self._ctx, byref(cu), byref(khz), byref(mem), name, 256))
        return {"cu_count": cu.value, "clock_khz": khz.value, "hbm_bytes": mem.value, "name": name.value.decode()}

    def set_stream(self, hip_stream):
        """Adopt an external hipStream_t given as an int handle, e.g.
        torch.cuda.current_stream().cuda_stream -- 0 is the device's default stream (torch's default).
        None goes back to the context's own stream."""
        if hip_stream is None:
            self._check(self._lib.ludvm_set_stream(self._ctx, None, 0))
        else:
            self._check(self._lib.ludvm_set_stream(self._ctx, c_void_p(int(hip_stream)), 1))

    def synchronize(self):
        self._check(self._lib.ludvm_synchronize(self._ctx))

    def set_tuning(self, targets_per_lane=0, source_splits=0):
        self._check(self._lib.ludvm_set_tuning(self._ctx, int(targets_per_lane), int(source_splits)))

    def set_sym_tuning(self, vortices_per_lane=0, rotation_split=0):
        """Tuning of the symmetric kernel (0 = heuristics): tile = 64 * vortices_per_lane (4 or 8), wavefronts
        sharing one tile pair's rotation steps (1, 2, 4)."""
        self._check(self._lib.ludvm_set_sym_tuning(self._ctx, int(vortices_per_lane), int(rotation_split)))

    def set_symmetric(self, mode=1):
        """0: always the direct kernel; 1: self-interaction launches may use the symmetric kernel (each unordered
        pair once, 64-bit fixed-point accumulation: also bitwise reproducible); n >= 2: the same from n vortices."""
        self._check(self._lib.ludvm_set_symmetric(self._ctx, int(mode)))

    def set_shard(self, rank, world, allreduce=None, d_acc=0, acc_bytes=0, min_vortices=0):
        """Evaluate only tile block `rank` of `world` of every symmetric roll-up of at least `min_vortices` vortices;
        `allreduce(count, stream)` must enqueue, on the hipStream_t `stream` (an int handle; 0 = the default stream), an
        in-place sum all-reduce of the first `count` int64 of the accumulator buffer at device address `d_acc`
        (ludvm_set_shard).  world = 1 undoes it."""
        if world > 1:
            def hook(_user, _buf, count, stream):
                try:
                    allreduce(int(count), int(stream or 0))
                    return 0
                except Exception as e:       # never let an exception cross the C frame
                    self._hook_error = e
                    return 1
            self._hook_c = _ffi.ALLREDUCE_FN(hook)        # kept alive as long as the library may call it
            self._check(self._lib.ludvm_set_shard(self._ctx, int(rank), int(world), int(min_vortices),
                                                  ctypes.cast(self._hook_c, c_void_p), None, c_void_p(int(d_acc)),
                                                  int(acc_bytes)))
        else:
            self._check(self._lib.ludvm_set_shard(self._ctx, 0, 1, 0, None, None, None, 0))
            self._hook_c = None

    # -- the library's own RCCL communicator -------------------------------------------------------
    def comm_unique_id(self):
        """128 bytes that identify a new communicator (rank 0 creates them and hands them to the other processes)."""
        _ffi.prefer_matching_rccl()
        buf = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
        rc = self._lib.ludvm_comm_unique_id(buf, _ffi.COMM_ID_BYTES)
        if rc != _ffi.OK:
            raise LudvmHipError(rc, "ludvm_comm_unique_id failed (is librccl loadable? LUDVM_RCCL_LIB names it)")
        return buf.raw

    def comm_init(self, rank, world, unique_id, min_vortices=0):
        """Join the communicator `unique_id` as rank `rank` of `world` (one process per GPU; returns when all have joined)
        and shard this engine's symmetric roll-ups over it: one in-library ncclAllReduce per time step."""
        _ffi.prefer_matching_rccl()
        uid = bytes(unique_id)
        if len(uid) != _ffi.COMM_ID_BYTES:
            raise ValueError("comm_init: the identifier is 128 bytes (Engine.comm_unique_id)")
        self._check(self._lib.ludvm_comm_init(self._ctx, int(rank), int(world), uid, len(uid), int(min_vortices)))

    @staticmethod
    def comm_init_all(engines, min_vortices=0):
        """ONE process, one engine per device: the engines join a new communicator in one call (ncclCommInitAll; engines[k]
        becomes rank k) and are sharded as by comm_init.  Each engine is then driven by a host thread of its own
        (ludvm_amd/multi.py)."""
        _ffi.prefer_matching_rccl()
        engines = list(engines)
        arr = (c_void_p * len(engines))(*[e._ctx for e in engines])
        engines[0]._check(engines[0]._lib.ludvm_comm_init_all(arr, len(engines), int(min_vortices)))

    def comm_destroy(self):
        self._check(self._lib.ludvm_comm_destroy(self._ctx))

    def comm_info(self):
        """(rank, world) of the engine's communicator; world = 0 when it has none."""
        r, w = c_int(), c_int()
        self._check(self._lib.ludvm_comm_info(self._ctx, byref(r), byref(w)))
        return r.value, w.value

    def comm_allreduce_i64_dev(self, d_buf, count):
        self._check(self._lib.ludvm_comm_allreduce_i64_dev(self._ctx, d_buf, int(count)))

    def comm_allgather_dev(self, d_send, d_recv, bytes_per_rank):
        self._check(self._lib.ludvm_comm_allgather_dev(self._ctx, d_send, d_recv, int(bytes_per_rank)))

    def comm_allgather(self, local):
        """All ranks' copies of the contiguous array `local` (same shape and dtype everywhere) stacked along a new first
        axis: [world, *local.shape] (host arrays; synchronous)."""
        local = np.ascontiguousarray(local)
        _, world = self.comm_info()
        out = np.empty((world,) + local.shape, local.dtype)
        self._check(self._lib.ludvm_comm_allgather_host(self._ctx, local.ctypes.data_as(c_void_p), out.ctypes.data_as(c_void_p),
                                                        local.nbytes))
        return out

    # -- stateless pair sum ----------------------------------------------------------------------
    def induce(self, circulation, xw, zw, xp, zp, v_core, precision="f32"):
        """(u, w) float64 arrays; host arrays in, host arrays out (LUDVM.py:549-570)."""
        g, xs, zs = _f64(circulation), _f64(xw), _f64(zw)
        # the same objects as targets (self-interaction) stay the same buffers, which lets the library
        # pick the symmetric kernel
        xt = xs if xp is xw else _f64(xp)
        zt = zs if zp is zw else _f64(zp)
        if not (len(g) == len(xs) == len(zs)) or len(xt) != len(zt):
            raise ValueError("induce: source arrays (and target arrays) must have equal lengths")
        u, w = np.empty(len(xt)), np.empty(len(xt))
        self._check(self._lib.ludvm_induce_f64(self._ctx, _pd(xs), _pd(zs), _pd(g), len(xs), _pd(xt), _pd(zt), len(xt),
                                               float(v_core), _prec(precision), _pd(u), _pd(w)))
        return u, w

    def spatial_order(self, x, z, with_extent=False):
        """(order, reordered[, mean_class_extent]): the order the library evaluates unordered points in -- Morton order, or
        the identity when the given order is already compact or there are fewer than 2048 points (ludvm_spatial_order).
        Position k of an array stored in that order holds the caller's element order[k].  mean_class_extent: how compact
        the 128-point origin classes are in that order (0.0 below 2048 points); fp32 on local origins keeps its tier up
        to about 150 v_core for a reordered cloud, 300 v_core for a set that is compact as given."""
        xs, zs = _f64(x), _f64(z)
        if len(xs) != len(zs):
            raise ValueError("x and z must have the same length")
        order = np.empty(len(xs), np.uint32)
        flag, ext = c_int(0), c_double(0.0)
        self._check(self._lib.ludvm_spatial_order(self._ctx, _pd(xs), _pd(zs), len(xs),
                                                  order.ctypes.data_as(POINTER(ctypes.c_uint)), byref(flag), byref(ext)))
        return (order, bool(flag.value), float(ext.value)) if with_extent else (order, bool(flag.value))

    def induce_f32(self, circulation, xw, zw, xp, zp, v_core):
        g, xs, zs, xt, zt = _f32(circulation), _f32(xw), _f32(zw), _f32(xp), _f32(zp)
        u, w = np.empty(len(xt), np.float32), np.empty(len(xt), np.float32)
        self._check(self._lib.ludvm_induce_f32(self._ctx, _pf(xs), _pf(zs), _pf(g), len(xs), _pf(xt), _pf(zt), len(xt),
                                               float(v_core), _pf(u), _pf(w)))
        return u, w

    def induce_dev(self, d_xs, d_zs, d_gs, ns, d_xt, d_zt, nt, v_core, d_u, d_w):
        """Raw device pointers (ints), fp32 SoA; asynchronous on the context stream."""
        self._check(self._lib.ludvm_induce_dev_f32(self._ctx, d_xs, d_zs, d_gs, ns, d_xt, d_zt, nt, float(v_core),
                                                   d_u, d_w))

    def advect_dev(self, d_xs, d_zs, d_gs, ns, t_first, nt, v_core, dt, d_x_out, d_z_out):
        self._check(self._lib.ludvm_advect_dev_f32(self._ctx, d_xs, d_zs, d_gs, ns, t_first, nt, float(v_core),
                                                   float(dt), d_x_out, d_z_out))

    def sym_scale_dev(self, d_g, n, v_core, d_scale):
        """Fixed-point scale of the symmetric kernel's raw sums for circulations d_g[n] -> the 32-byte device record
        d_scale (the same bits on every GPU that holds the same circulations)."""
        self._check(self._lib.ludvm_sym_scale_dev_f32(self._ctx, d_g, n, float(v_core), d_scale))

    def sym_accumulate_dev(self, d_x, d_z, d_g, n, tile_first, tile_count, v_core, d_scale, d_acc_u, d_acc_w, d_bad):
        """This owner's share of the unordered pairs, ADDED as 64-bit fixed-point integers into d_acc_u / d_acc_w
        (int64[n], zeroed by the caller); d_bad (int64[1]) counts non-finite partial sums."""
        self._check(self._lib.ludvm_sym_accumulate_dev_f32(self._ctx, d_x, d_z, d_g, n, tile_first, tile_count,
                                                           float(v_core), d_scale, d_acc_u, d_acc_w, d_bad))

    def advect_from_sums_dev(self, d_sum_u, d_sum_w, d_scale, d_bad, d_x, d_z, t_first, nt, dt, d_x_out, d_z_out):
        self._check(self._lib.ludvm_advect_from_sums_dev_f32(self._ctx, d_sum_u, d_sum_w, d_scale, d_bad, d_x, d_z,
                                                             t_first, nt, float(dt), d_x_out, d_z_out))

    # -- resident wake ---------------------------------------------------------------------------
    def wake_reserve(self, capacity):
        self._check(self._lib.ludvm_wake_reserve(self._ctx, int(capacity)))

    def wake_clear(self):
        self._check(self._lib.ludvm_wake_clear(self._ctx))

    def wake_size(self):
        n = c_size_t()
        self._check(self._lib.ludvm_wake_size(self._ctx, byref(n)))
        return n.value

    def wake_truncate(self, n):
        self._check(self._lib.ludvm_wake_truncate(self._ctx, int(n)))

    def wake_append(self, x, z, gamma):
        x, z, g = _f64(x), _f64(z), _f64(gamma)
        if not (len(x) == len(z) == len(g)):
            raise ValueError("wake_append: x, z, gamma must have equal lengths")
        self._check(self._lib.ludvm_wake_append(self._ctx, _pd(x), _pd(z), _pd(g), len(x)))

    def wake_write(self, first, x=None, z=None, gamma=None):
        arrs = [None if a is None else _f64(a) for a in (x, z, gamma)]
        lens = {len(a) for a in arrs if a is not None}
        if len(lens) != 1:
            raise ValueError("wake_write: give at least one field; all given fields must have equal lengths")
        self._check(self._lib.ludvm_wake_write(self._ctx, int(first), lens.pop(), _pd(arrs[0]), _pd(arrs[1]),
                                               _pd(arrs[2])))

    def wake_read(self, first, count, gamma=False):
        x, z = np.empty(count), np.empty(count)
        g = np.empty(count) if gamma else None
        self._check(self._lib.ludvm_wake_read(self._ctx, int(first), int(count), _pd(x), _pd(z), _pd(g)))
        return (x, z, g) if gamma else (x, z)

    def wake_induce_on_points(self, src_first, src_count, xp, zp, v_core):
        xt, zt = _f64(xp), _f64(zp)
        u, w = np.empty(len(xt)), np.empty(len(xt))
        self._check(self._lib.ludvm_wake_induce_on_points(self._ctx, int(src_first), int(src_count), _pd(xt), _pd(zt),
                                                          len(xt), float(v_core), _pd(u), _pd(w)))
        return u, w

    def wake_chord_sums(self, src_first, src_count, xp, zp, unit_x, unit_z, v_core):
        """(u_wake, w_wake)[nt] from the resident wake and (u_unit, w_unit)[n_unit, nt] from unit
        vortices at (unit_x, unit_z), in one round trip (fp64)."""
        xt, zt, ux, uz = _f64(xp), _f64(zp), _f64(unit_x), _f64(unit_z)
        nt, nu = len(xt), len(ux)
        u, w = np.empty(nt), np.empty(nt)
        uu, wu = np.empty([nu, nt]), np.empty([nu, nt])
        self._check(self._lib.ludvm_wake_chord_sums(self._ctx, int(src_first), int(src_count), _pd(xt), _pd(zt), nt,
                                                    _pd(ux), _pd(uz), nu, float(v_core), _pd(u), _pd(w), _pd(uu), _pd(wu)))
        return u, w, uu, wu

    def step_buffers(self, npoints):
        """Preallocated host buffers (and their ctypes pointers) for the two per-step calls of a time
        loop with `npoints` chord points: removes the per-call array allocation and pointer casting."""
        b = type("StepBuffers", (), {})()
        b.n = npoints
        b.unit = np.zeros([2, 2])                       # [x|z][tev, lev]
        b.u, b.w = np.empty(npoints), np.empty(npoints)
        b.uu, b.wu = np.empty([2, npoints]), np.empty([2, npoints])
        b.tail = np.empty([2, 2])                       # [x|z][newest two]
        b.p_unit_x, b.p_unit_z = _pd(b.unit[0]), _pd(b.unit[1])
        b.p_u, b.p_w, b.p_uu, b.p_wu = _pd(b.u), _pd(b.w), _pd(b.uu), _pd(b.wu)
        b.p_tx, b.p_tz = _pd(b.tail[0]), _pd(b.tail[1])
        return b

    def wake_chord_sums_into(self, b, src_count, xp, zp, v_core):
        """wake_chord_sums over sources [0, src_count) with the unit vortices in b.unit, results in
        b.u, b.w, b.uu, b.wu.  xp, zp must be contiguous float64 arrays of length b.n."""
        self._check(self._lib.ludvm_wake_chord_sums(self._ctx, 0, src_count, _pd(xp), _pd(zp), b.n, b.p_unit_x, b.p_unit_z,
                                                    2, v_core, b.p_u, b.p_w, b.p_uu, b.p_wu))

    def wake_advect_tail_into(self, b, dt, foil_x, foil_z, foil_dgamma, v_core, tail_count, precision):
        """wake_advect_tail with contiguous float64 foil arrays; newest positions land in b.tail[:, :tail_count]."""
        self._check(self._lib.ludvm_wake_advect_tail(self._ctx, dt, _pd(foil_x), _pd(foil_z), _pd(foil_dgamma),
                                                     len(foil_x), v_core, precision, tail_count, b.p_tx, b.p_tz))

    def wake_step_into(self, b, new_x, new_z, new_gamma, dt, foil_x, foil_z, foil_dgamma, v_core, precision, te, le,
                       lev_from_prev, tail_count, xp_next, zp_next):
        """Append this step's shed vortices, roll the wake up, and return the placement and chord sums
        of the next step -- one packed upload and one download (ludvm_wake_step).  Results:
        b.tail[:, :tail_count], b.unit (placed TEV / LEV candidate), b.u, b.w, b.uu, b.wu for the next
        step.  All array arguments contiguous float64."""
        self._check(self._lib.ludvm_wake_step(self._ctx, _pd(new_x), _pd(new_z), _pd(new_gamma), len(new_x), dt,
                                              _pd(foil_x), _pd(foil_z), _pd(foil_dgamma), len(foil_x), v_core, precision,
                                              _pd(te), _pd(le), 1 if lev_from_prev else 0, tail_count, _pd(xp_next),
                                              _pd(zp_next), b.n, b.p_tx, b.p_tz, b.p_unit_x, b.p_unit_z, b.p_u, b.p_w,
                                              b.p_uu, b.p_wu))

    def wake_advect_tail(self, dt, foil_x, foil_z, foil_dgamma, v_core, tail_count, precision="f32"):
        """Roll-up step, then the updated (x, z) of the last `tail_count` wake vortices."""
        fx, fz, fg = _f64(foil_x), _f64(foil_z), _f64(foil_dgamma)
        tx, tz = np.empty(tail_count), np.empty(tail_count)
        self._check(self._lib.ludvm_wake_advect_tail(self._ctx, float(dt), _pd(fx), _pd(fz), _pd(fg), len(fx),
                                                     float(v_core), _prec(precision), int(tail_count), _pd(tx), _pd(tz)))
        return tx, tz

    def wake_advect(self, dt, foil_x, foil_z, foil_dgamma, v_core, precision="f32", return_velocity=False):
        fx, fz, fg = _f64(foil_x), _f64(foil_z), _f64(foil_dgamma)
        u = w = None
        if return_velocity:
            n = self.wake_size()
            u, w = np.empty(n), np.empty(n)
        self._check(self._lib.ludvm_wake_advect(self._ctx, float(dt), _pd(fx), _pd(fz), _pd(fg), len(fx),
                                                float(v_core), _prec(precision), _pd(u), _pd(w)))
        return (u, w) if return_velocity else None

    # -- device-resident time march ----------------------------------------------------------------
    MARCH_ROW_HEAD = 12
    MARCH_STATE_HEAD = 16

    def march_setup(self, npan, ncoef, scalars, tables, kin):
        """Upload what a run keeps constant (ludvm_march_setup): scalars [Uinf, chord, rho, dt, piv, v_core, IC,
        sum(Gamma_free), method (0 Faure / 1 Ramesh), maxerror, maxiter, epsilon], the packed chord tables and the
        per-step kinematics rows [nt, 7 + 2 npan]."""
        sc, tb = _f64(scalars), _f64(tables)
        kin = np.ascontiguousarray(kin, dtype=np.float64)
        if kin.ndim != 2 or kin.shape[1] != 7 + 2 * npan or len(sc) != 12:
            raise ValueError("march_setup: kin must be [nt, 7 + 2 npan] and scalars 12 long")
        if len(tb) != 8 * npan + ncoef * npan + (ncoef - 1) * npan:
            raise ValueError("march_setup: wrong table length")
        self._check(self._lib.ludvm_march_setup(self._ctx, int(npan), int(ncoef), _pd(sc), _pd(tb), _pd(kin), kin.shape[0]))
        self._march_dims = (int(npan), int(ncoef))

    def march_run(self, first_step, count, precision, state, hist_nmax=0, anchors=None):
        """Advance the resident wake through time steps [first_step, first_step + count) without a host round
        trip per step (ludvm_march_run).  `state` (16 + ncoef float64) is updated in place; returns the
        per-step rows [count, 12 + 2 ncoef + 2 npan] -- and, with hist_nmax > 0, the positions of every wake
        vortex after each step, [count, 2, hist_nmax] (the reference's dense history).  `anchors`: the wake sizes after
        the four anchor steps `march_anchor_steps(first_step)` (-1 = not given; None = none), which make the launch
        geometry -- and with it the fp32 rounding -- independent of where the calls begin."""
        npan, ncoef = self._march_dims
        if state.dtype != np.float64 or not state.flags.c_contiguous or len(state) != self.MARCH_STATE_HEAD + ncoef:
            raise ValueError("march_run: state must be contiguous float64 of length 16 + ncoef")
        rows = np.empty([int(count), self.MARCH_ROW_HEAD + 2 * ncoef + 2 * npan])
        hist = np.empty([int(count), 2, int(hist_nmax)]) if hist_nmax else None
        anc = None
        if anchors is not None:
            anc = (ctypes.c_longlong * 4)(*[int(v) for v in anchors])
        self._check(self._lib.ludvm_march_run(self._ctx, int(first_step), int(count), _prec(precision), _pd(state), _pd(rows),
                                              _pd(hist), int(hist_nmax), anc))
        return (rows, hist) if hist_nmax else rows

    @staticmethod
    def march_anchor_steps(first_step):
        """The four steps whose wake sizes ludvm_march_run wants in `anchors` for a call that begins at first_step."""
        return [max(64 * (int(first_step) // 64 - 3 + q) - 1, 0) for q in range(4)]

    # -- flow field ------------------------------------------------------------------------------
    def flowfield(self, xmin, zmin, dr, nx, nz, circulation, xw, zw, v_core):
        """(u, w) float32 [nx, nz] on the grid (xmin + i*dr, zmin + j*dr) (LUDVM.py:1193-1195)."""
        g, xs, zs = _f64(circulation), _f64(xw), _f64(zw)
        u, w = np.empty(nx * nz, np.float32), np.empty(nx * nz, np.float32)
        self._check(self._lib.ludvm_flowfield_f32(self._ctx, float(xmin), float(zmin), float(dr), int(nx), int(nz),
                                                  _pd(xs), _pd(zs), _pd(g), len(xs), float(v_core), _pf(u), _pf(w)))
        return u.reshape(nx, nz), w.reshape(nx, nz)

    def flowfield_vorticity(self, xmin, zmin, dr, nx, nz, circulation, xw, zw, v_core, precision="f32"):
        """(u, w, ome) [nx, nz]: velocity field and its vorticity (LUDVM.py:1224-1292) in one device round trip -- the
        stencil runs on the fields where they are.  float32 arrays from fp32 arithmetic on local-origin sources, or,
        with precision='f64', float64 arrays from float64 arithmetic throughout (the reference's)."""
        return self.flowfield_rows(xmin, zmin, dr, nx, nz, 0, nx, circulation, xw, zw, v_core, precision=precision)

    def flowfield_rows(self, xmin, zmin, dr, nx, nz, row_first, row_count, circulation, xw, zw, v_core, vorticity=True,
                       precision="f32"):
        """Rows [row_first, row_first + row_count) of the nx x nz grid: (u, w, ome) [row_count, nz], bit for bit what the
        whole-grid call returns for those rows -- the unit a multi-GPU flow field shards by.  precision 'f32' (float32
        arrays) or 'f64' (float64 arrays, float64 arithmetic throughout)."""
        g, xs, zs = _f64(circulation), _f64(xw), _f64(zw)
        f64 = _prec(precision) == PREC_F64
        dt, ptr = (np.float64, _pd) if f64 else (np.float32, _pf)
        fn = self._lib.ludvm_flowfield_rows_f64 if f64 else self._lib.ludvm_flowfield_rows_f32
        u, w = np.empty(row_count * nz, dt), np.empty(row_count * nz, dt)
        ome = np.empty(row_count * nz, dt) if vorticity else None
        self._check(fn(self._ctx, float(xmin), float(zmin), float(dr), int(nx), int(nz), int(row_first), int(row_count),
                       _pd(xs), _pd(zs), _pd(g), len(xs), float(v_core), ptr(u), ptr(w), ptr(ome)))
        sh = (row_count, nz)
        return u.reshape(sh), w.reshape(sh), (ome.reshape(sh) if vorticity else None)

    def flowfield_dev(self, xmin, zmin, dr, nx, nz, d_xs, d_zs, d_gs, ns, v_core, d_u, d_w):
        self._check(self._lib.ludvm_flowfield_dev_f32(self._ctx, float(xmin), float(zmin), float(dr), int(nx), int(nz),
                                                      d_xs, d_zs, d_gs, int(ns), float(v_core), d_u, d_w))

    def vorticity(self, u, w, dr):
        nx, nz = u.shape
        uu, ww = _f32(u), _f32(w)
        ome = np.empty(nx * nz, np.float32)
        self._check(self._lib.ludvm_vorticity_f32(self._ctx, _pf(uu), _pf(ww), nx, nz, float(dr), _pf(ome)))
        return ome.reshape(nx, nz)

    def vorticity_dev(self, d_u, d_w, nx, nz, dr, d_ome):
        self._check(self._lib.ludvm_vorticity_dev_f32(self._ctx, d_u, d_w, int(nx), int(nz), float(dr), d_ome))

    def fixed_point_probe(self, values, scale_log2):
        """int64 units the symmetric kernel adds to an accumulator for fp32 partial sums `values` at scale
        2**scale_log2 (ludvm_fixed_point_probe): trunc(value * scale), exact below 2**63."""
        v = _f32(values)
        out = np.empty(len(v), np.int64)
        self._check(self._lib.ludvm_fixed_point_probe(self._ctx, _pf(v), len(v), int(scale_log2),
                                                      out.ctypes.data_as(POINTER(c_longlong))))
        return out

    # -- measurement -----------------------------------------------------------------------------
    def kernel_timing(self, enable=True):
        self._check(self._lib.ludvm_kernel_timing(self._ctx, 1 if enable else 0))

    def kernel_time_ms(self, reset=True):
        ms, n = c_double(), c_longlong()
        self._check(self._lib.ludvm_kernel_time_ms(self._ctx, 1 if reset else 0, byref(ms), byref(n)))
        return ms.value, n.value
