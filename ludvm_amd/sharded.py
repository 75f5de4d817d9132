"""Multi-GPU wake self-advection: the pair work is sharded across ranks and the updated source
positions are republished with one all-gather per step (BASELINE config 4).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank holds
the full source SoA (x, z, Gamma: 12 B per vortex, 96 MB at N = 8e6) and owns the contiguous block
of vortices [rank*n_loc, (rank+1)*n_loc).  Two step variants:

direct (symmetric=False) -- what the north star describes:
    1. pair kernel: all N sources -> own n_loc targets, fused explicit-Euler update of the own block
       (reference LUDVM.py:1105-1109 with the wake as both source and target set), written straight
       into this rank's send slot [2, n_loc] (x row, z row);
    2. ONE all-gather of the [2, n_loc] blocks -> [G, 2, n_loc]; one strided device copy lays them out
       as the next contiguous x[N], z[N] (Gamma never moves).

symmetric (default) -- each unordered pair evaluated once, chip-wide 1.5x faster:
    1. symmetric kernel over the rank's own I-tiles of the global tile ring: J = I + d (mod NT),
       d <= NT/2, so a rank's work depends only on how many tiles it owns (exactly 1/G of the job) and
       touches its own block plus the next half of the ring; raw (u, w) sums of BOTH partners are
       accumulated, as 64-bit fixed-point integers, into a full-length buffer [2, N] (+ one NaN counter);
    2. ONE all-reduce (integer sum) of that buffer (16 B per vortex over xGMI, ~1-2 ms at N = 8e6 against
       ~1 s of pair arithmetic).  Integer addition is associative, so every rank ends up with the same
       bits -- the bits one GPU owning all tiles would have produced -- whatever the ring order of the
       collective;
    3. every rank Euler-updates ALL N vortices from the complete sums (O(N), replicated): no position
       exchange is needed, and the replicas cannot drift apart.

The pair arithmetic is injected (`kernel`): the product passes HipShardKernel (HIP engine, device
pointers); the CPU tests pass a checker built on the oracle to cover the sharding / collective logic
under gloo.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from ._ffi import SYM_OWNER_ALIGN, SYM_TILE

PAD_POS = 1.0e6  # padding vortices (block sizes are rounded up): zero strength, far away


class HipShardKernel:
    """The pair arithmetic of a shard step on CUDA/HIP float32 tensors, via the engine's
    device-pointer entry points (asynchronous on torch's current stream)."""

    def __init__(self, engine):
        self.engine = engine

    def _check(self, *tensors):
        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("HipShardKernel needs contiguous float32 device tensors")
        self.engine.set_stream(torch.cuda.current_stream().cuda_stream)

    def advect(self, xs, zs, gs, t_first, nt, v_core, dt, x_out, z_out):
        self._check(xs, zs, gs, x_out, z_out)
        self.engine.advect_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), xs.numel(), t_first, nt, v_core, dt,
                               x_out.data_ptr(), z_out.data_ptr())

    def sym_scale(self, gs, v_core, scale):
        """scale: uint8[32] device tensor receiving the fixed-point scale record for circulations gs."""
        self._check(gs)
        self.engine.sym_scale_dev(gs.data_ptr(), gs.numel(), v_core, scale.data_ptr())

    def sym_accumulate(self, xs, zs, gs, tile_first, tile_count, v_core, scale, acc):
        """acc: int64[2 * n + 1] = raw u sums | raw w sums | NaN counter (zeroed by the caller)."""
        self._check(xs, zs, gs)
        n = xs.numel()
        base = acc.data_ptr()
        self.engine.sym_accumulate_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), n, tile_first, tile_count, v_core,
                                       scale.data_ptr(), base, base + 8 * n, base + 16 * n)

    def advect_from_sums(self, acc, scale, xs, zs, t_first, nt, dt, x_out, z_out):
        self._check(xs, zs, x_out, z_out)
        n = xs.numel()
        base = acc.data_ptr()
        self.engine.advect_from_sums_dev(base + 8 * t_first, base + 8 * (n + t_first), scale.data_ptr(), base + 16 * n,
                                         xs.data_ptr(), zs.data_ptr(), t_first, nt, dt, x_out.data_ptr(), z_out.data_ptr())


def flowfield_rows(nx, world, rank):
    """Contiguous block of grid rows (x index) owned by `rank`: [r0, r1)."""
    per = (nx + world - 1) // world
    r0 = min(nx, rank * per)
    return r0, min(nx, r0 + per)


class ShardedFlowfield:
    """LUDVM.flowfield (LUDVM.py:1186-1298) with the grid rows sharded across ranks: every rank holds
    all sources and evaluates rows [r0, r1) of the x-major grid (plus one halo row on each interior
    side, so that the vorticity stencil needs no exchange).  There is no collective on the data path;
    `gather()` assembles the full fields on every rank for callers that want them.

    `kernel` provides flowfield(xmin, zmin, dr, nx, nz, xs, zs, gs, v_core, u, w) and
    vorticity(u, w, nx, nz, dr, ome) on [nx, nz] float32 tensors (HipFlowfieldKernel for the product)."""

    def __init__(self, kernel, device, group=None):
        self.kernel, self.device, self.group = kernel, device, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def compute(self, xmin, zmin, dr, nx, nz, xs, zs, gs, v_core):
        """(u, w, ome) float32 tensors [r1 - r0, nz] for this rank's rows."""
        r0, r1 = flowfield_rows(nx, self.world, self.rank)
        self.rows = (r0, r1)
        if r1 <= r0:
            e = torch.empty([0, nz], dtype=torch.float32, device=self.device)
            return e, e.clone(), e.clone()
        h0, h1 = max(0, r0 - 1), min(nx, r1 + 1)          # halo rows for the centred differences
        n_rows = h1 - h0
        u = torch.empty([n_rows, nz], dtype=torch.float32, device=self.device)
        w = torch.empty_like(u)
        ome = torch.empty_like(u)
        self.kernel.flowfield(xmin + h0 * dr, zmin, dr, n_rows, nz, xs, zs, gs, v_core, u, w)
        self.kernel.vorticity(u, w, n_rows, nz, dr, ome)
        # rows computed with a one-sided difference only because the halo ended there are dropped; rows
        # at the true grid edge keep the reference's one-sided form (:1233-1248)
        lo, hi = r0 - h0, r0 - h0 + (r1 - r0)
        return u[lo:hi], w[lo:hi], ome[lo:hi]

    def gather(self, field, nx):
        """All ranks' row blocks of `field` stacked into [nx, nz] (blocks are padded to equal height)."""
        if self.world == 1:
            return field
        per = (nx + self.world - 1) // self.world
        nz = field.shape[1]
        send = torch.zeros([per, nz], dtype=field.dtype, device=field.device)
        send[: field.shape[0]] = field
        recv = torch.empty([self.world * per, nz], dtype=field.dtype, device=field.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        return recv[:nx]


class HipFlowfieldKernel:
    def __init__(self, engine):
        self.engine = engine

    def flowfield(self, xmin, zmin, dr, nx, nz, xs, zs, gs, v_core, u, w):
        self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.engine.flowfield_dev(xmin, zmin, dr, nx, nz, xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), xs.numel(), v_core,
                                  u.data_ptr(), w.data_ptr())

    def vorticity(self, u, w, nx, nz, dr, ome):
        self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.engine.vorticity_dev(u.data_ptr(), w.data_ptr(), nx, nz, dr, ome.data_ptr())


class ShardedWake:
    def __init__(self, x, z, gamma, v_core, dt, kernel, device, group=None, symmetric=True, force_collectives=False,
                 collectives="torch"):
        # force_collectives: issue the collectives even in a one-rank group (they are then identities); lets a
        # one-GPU box run the real RCCL calls with the layouts used at G > 1
        # collectives: "torch" -- torch.distributed on `group`; "library" -- the engine's own RCCL communicator
        # (ludvm_comm_init was called on kernel.engine): ncclAllReduce / ncclAllGather issued inside libludvm_hip.so on the
        # stream the pair kernel runs on, rank and world taken from that communicator
        self.force = bool(force_collectives)
        self.group = group
        self.library = collectives == "library"
        if collectives not in ("torch", "library"):
            raise ValueError("collectives must be 'torch' or 'library'")
        if self.library:
            self.rank, self.world = kernel.engine.comm_info()
            if self.world < 1:
                raise RuntimeError("collectives='library' needs Engine.comm_init first")
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.kernel, self.device = kernel, device
        self.v_core, self.dt = float(v_core), float(dt)
        self.symmetric = bool(symmetric)
        self.n = len(x)
        g = self.world
        n_loc = (self.n + g - 1) // g
        if self.symmetric:   # block boundaries must fall on quad boundaries (4 tiles) of the symmetric kernel
            unit = SYM_TILE * SYM_OWNER_ALIGN
            n_loc = (n_loc + unit - 1) // unit * unit
        self.n_loc = n_loc
        n_pad = n_loc * g
        pad = n_pad - self.n

        def padded(a, fill):
            a = np.asarray(a, dtype=np.float32)
            return torch.from_numpy(np.concatenate([a, np.full(pad, fill, np.float32)])).to(device)

        self.xs, self.zs, self.gs = padded(x, PAD_POS), padded(z, PAD_POS), padded(gamma, 0.0)
        self.n_pad = n_pad
        self.lo = self.rank * n_loc
        self._send = torch.empty([2, n_loc], dtype=torch.float32, device=device)
        self._recv = torch.empty([g, 2, n_loc], dtype=torch.float32, device=device)
        self._xz = torch.empty([2, n_pad], dtype=torch.float32, device=device)
        if self.symmetric:
            # raw sums of all vortices as 64-bit fixed point: u[n_pad] | w[n_pad] | NaN counter
            self._acc = torch.zeros([2 * n_pad + 1], dtype=torch.int64, device=device)
            self._scale = torch.zeros([32], dtype=torch.uint8, device=device)
            self.kernel.sym_scale(self.gs, self.v_core, self._scale)     # circulations do not change: once
        self._timing = False
        self._coll_events, self._coll_ms, self._coll_count = [], 0.0, 0

    # ---- the collective's own time ----------------------------------------------------------------------------------------
    def collective_timing(self, enable):
        """Time every step's collective from here on: events on the stream it is issued on (torch's current stream, which
        the engine launches on too), so the figure is what the step waits between the pair kernel and the Euler update --
        the transfer AND the wait for the slowest rank's kernel."""
        self._timing = bool(enable)

    def _coll_begin(self):
        if not self._timing:
            return None
        if self.device.type != "cuda":
            return time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def _coll_end(self, tok):
        if tok is None:
            return
        if self.device.type != "cuda":
            self._coll_ms += (time.perf_counter() - tok) * 1e3
            self._coll_count += 1
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self._coll_events.append((tok, e1))
        if len(self._coll_events) >= 1024:          # (a caller that never asks: fold the finished pairs in, keep the list short)
            self.collective_time_ms(reset=False)

    def collective_time_ms(self, reset=True):
        """(average ms per collective, collectives timed) since the last reset; waits for the recorded events."""
        for e0, e1 in self._coll_events:
            e1.synchronize()
            self._coll_ms += e0.elapsed_time(e1)
            self._coll_count += 1
        self._coll_events = []
        avg, cnt = (self._coll_ms / self._coll_count if self._coll_count else 0.0), self._coll_count
        if reset:
            self._coll_ms, self._coll_count = 0.0, 0
        return avg, cnt

    # ---- do the replicas hold the same wake? ------------------------------------------------------------------------------
    def checksum(self):
        """Four 64-bit integers over the bit patterns of the current positions of the N real vortices: plain and
        index-weighted sums of x and of z (the weights, 1 .. 251 by index, catch a permutation; |bits| < 2^31, so the sums
        stay below 2^63 up to 2^23 = 8.4e6 vortices and wrap -- identically on every rank -- beyond).  Every rank of a
        sharded run must report the same four numbers."""
        idx = torch.arange(self.n, device=self.xs.device, dtype=torch.int64) % 251 + 1
        out = []
        for a in (self.xs, self.zs):
            bits = a[: self.n].contiguous().view(torch.int32).to(torch.int64)
            out += [int(bits.sum().item()), int((bits * idx).sum().item())]
        return out

    def gather_checksums(self):
        """[world][4] checksums of all ranks (on every rank), through the same channel the step's collective uses."""
        mine = self.checksum()
        if self.world == 1:
            return [mine]
        if self.library:
            allc = self.kernel.engine.comm_allgather(np.asarray(mine, dtype=np.int64))
            return [[int(v) for v in row] for row in np.asarray(allc).reshape(self.world, 4)]
        dev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        send = torch.tensor(mine, dtype=torch.int64, device=dev)
        recv = torch.empty([self.world * 4], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return [[int(v) for v in row] for row in recv.view(self.world, 4).cpu()]

    def ranks_agree(self):
        sums = self.gather_checksums()
        return all(s == sums[0] for s in sums)

    @property
    def pairs_per_step(self):
        """Whole-job ordered pair interactions of one step (self pairs count; padding does not)."""
        return float(self.n) * float(self.n)

    def step(self):
        if self.symmetric:
            # own I-tiles of the global tile ring -> raw sums of both partners; ONE integer all-reduce; every rank
            # then moves every vortex itself (identical bits everywhere: no position exchange)
            self._acc.zero_()
            tiles = self.n_loc // SYM_TILE
            self.kernel.sym_accumulate(self.xs, self.zs, self.gs, self.rank * tiles, tiles, self.v_core, self._scale,
                                       self._acc)
            if self.world > 1 or self.force:
                tok = self._coll_begin()
                if self.library:
                    self.kernel.engine.comm_allreduce_i64_dev(self._acc.data_ptr(), self._acc.numel())
                else:
                    dist.all_reduce(self._acc, op=dist.ReduceOp.SUM, group=self.group)
                self._coll_end(tok)
            nxt = self._xz if self.xs.data_ptr() != self._xz.data_ptr() else self._xz2()
            self.kernel.advect_from_sums(self._acc, self._scale, self.xs, self.zs, 0, self.n_pad, self.dt, nxt[0], nxt[1])
            self.xs, self.zs = nxt[0], nxt[1]
            return
        send = self._send
        self.kernel.advect(self.xs, self.zs, self.gs, self.lo, self.n_loc, self.v_core, self.dt, send[0], send[1])
        if self.world > 1 or self.force:
            tok = self._coll_begin()
            if self.library:
                self.kernel.engine.comm_allgather_dev(send.data_ptr(), self._recv.data_ptr(), send.numel() * 4)
            else:
                dist.all_gather_into_tensor(self._recv.view(-1), send.view(-1), group=self.group)
            self._coll_end(tok)
            # [G, 2, n_loc] -> [2, G*n_loc].  In-place reuse of _xz is safe: in stream order the pair
            # kernel that read it has finished before this copy starts.
            self._xz.view(2, self.world, self.n_loc).copy_(self._recv.permute(1, 0, 2))
            self.xs, self.zs = self._xz[0], self._xz[1]
        else:
            self._xz[:, : self.n_loc].copy_(send)
            self.xs, self.zs = self._xz[0], self._xz[1]

    def _xz2(self):
        if not hasattr(self, "_xz_b"):
            self._xz_b = torch.empty_like(self._xz)
        return self._xz_b

    def positions(self):
        """Current (x, z) of the N real vortices as float32 numpy arrays."""
        return self.xs[: self.n].cpu().numpy(), self.zs[: self.n].cpu().numpy()
