"""Multi-GPU wake self-advection: targets sharded across ranks, one all-gather of the updated
source positions per step (BASELINE config 4).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank holds
the full source SoA (x, z, Gamma: 12 B per vortex, 96 MB at N = 8e6) and owns the contiguous block
of targets [rank*n_loc, (rank+1)*n_loc).  A step is

    1. pair kernel: all N sources -> own n_loc targets, fused explicit-Euler update of the own block
       (reference LUDVM.py:1105-1109 with the wake as both source and target set), written straight
       into this rank's slot of the send buffer [2, n_loc] (x row, z row);
    2. ONE all-gather of the [2, n_loc] blocks -> [G, 2, n_loc];
    3. one strided device copy that lays the gathered blocks out as the next contiguous x[N], z[N]
       (Gamma never moves).

There is no other collective on the data path.  The pair arithmetic is injected (`kernel`): the
product passes HipShardKernel (HIP engine, device pointers); the CPU tests pass a checker built on
the oracle to cover the sharding / all-gather logic under gloo.
"""
import numpy as np
import torch
import torch.distributed as dist

PAD_POS = 1.0e6  # padding vortices (N not divisible by the world size): zero strength, far away


class HipShardKernel:
    """advect(xs, zs, gs, t_first, nt, v_core, dt, x_out, z_out) on CUDA/HIP tensors via the engine."""

    def __init__(self, engine):
        self.engine = engine

    def advect(self, xs, zs, gs, t_first, nt, v_core, dt, x_out, z_out):
        for t in (xs, zs, gs, x_out, z_out):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("HipShardKernel needs contiguous float32 device tensors")
        self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.engine.advect_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), xs.numel(), t_first, nt, v_core, dt,
                               x_out.data_ptr(), z_out.data_ptr())


class ShardedWake:
    def __init__(self, x, z, gamma, v_core, dt, kernel, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.kernel, self.device = kernel, device
        self.v_core, self.dt = float(v_core), float(dt)
        self.n = len(x)
        g = self.world
        self.n_loc = (self.n + g - 1) // g
        n_pad = self.n_loc * g
        pad = n_pad - self.n

        def padded(a, fill):
            a = np.asarray(a, dtype=np.float32)
            return torch.from_numpy(np.concatenate([a, np.full(pad, fill, np.float32)])).to(device)

        self.xs, self.zs, self.gs = padded(x, PAD_POS), padded(z, PAD_POS), padded(gamma, 0.0)
        self.n_pad = n_pad
        self.lo = self.rank * self.n_loc
        self._send = torch.empty([2, self.n_loc], dtype=torch.float32, device=device)
        self._recv = torch.empty([g, 2, self.n_loc], dtype=torch.float32, device=device)
        self._xz = torch.empty([2, n_pad], dtype=torch.float32, device=device)

    @property
    def pairs_per_step(self):
        """Whole-job ordered pair interactions of one step (self pairs count; padding does not)."""
        return float(self.n) * float(self.n)

    def step(self):
        send = self._send
        self.kernel.advect(self.xs, self.zs, self.gs, self.lo, self.n_loc, self.v_core, self.dt, send[0], send[1])
        if self.world > 1:
            dist.all_gather_into_tensor(self._recv.view(-1), send.view(-1), group=self.group)
            # [G, 2, n_loc] -> [2, G*n_loc].  In-place reuse of _xz is safe: in stream order the pair
            # kernel that read it has finished before this copy starts.
            self._xz.view(2, self.world, self.n_loc).copy_(self._recv.permute(1, 0, 2))
            self.xs, self.zs = self._xz[0], self._xz[1]
        else:
            self.xs, self.zs = send[0].clone(), send[1].clone()

    def positions(self):
        """Current (x, z) of the N real vortices as float32 numpy arrays."""
        return self.xs[: self.n].cpu().numpy(), self.zs[: self.n].cpu().numpy()
