"""One LUDVM simulation on several GPUs, behind the reference's own method surface.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank constructs the SAME
`LUDVM(..., distributed=True)` and calls the same methods; what is sharded, and what each method exchanges:

  LUDVM.flowfield          grid rows in contiguous blocks (ludvm_flowfield_rows_f32: each block bit for bit what the
    (LUDVM.py:1186-1298)   single-GPU call computes for those rows, halo rows for the vorticity stencil evaluated
                           internally); one all-gather of the finished (u, w, omega) rows per requested time step, so
                           that every rank ends up with the reference's full u_ff / w_ff / ome_ff arrays.
  LUDVM.induced_velocity   targets in contiguous blocks (sources replicated: every rank was handed them), one all-gather
    (LUDVM.py:549-570)     of the (u, w) blocks; calls with fewer than `min_targets` targets or `min_pairs` pairs stay on the calling GPU.
  LUDVM.time_loop          every rank holds the whole wake and runs the whole loop, but evaluates only its tile block of
    (LUDVM.py:1095-1127)   the symmetric roll-up kernel's unordered pairs; ONE integer all-reduce of the fixed-point sums
                           per time step (ludvm_set_shard), from `min_wake` vortices on.  Integer sums commute: every
                           rank reads the same bits, so the replicated state cannot drift, and the results equal the
                           single-GPU run bit for bit.

PyTorch is plumbing here: device memory for the accumulators and the collectives.  The pair arithmetic is the engine's.
"""
import numpy as np
import torch
import torch.distributed as dist

from .comm import MIN_PAIRS, MIN_TARGETS, MIN_WAKE


class ShardGroup:
    """The ranks that share one simulation (a torch.distributed process group; None = the default group)."""

    def __init__(self, group=None, device=None, min_targets=MIN_TARGETS, min_wake=MIN_WAKE, min_pairs=MIN_PAIRS):
        if not dist.is_initialized():
            raise RuntimeError("LUDVM(distributed=...) needs torch.distributed to be initialised (one process per GPU, "
                               "e.g. torchrun; backend 'nccl' = RCCL)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        # RCCL moves device tensors; gloo (CPU tests, one-GPU rehearsals) host tensors
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                         if self.backend == "nccl" else torch.device("cpu"))
        self.min_targets, self.min_wake, self.min_pairs = int(min_targets), int(min_wake), int(min_pairs)
        self._acc = None

    # ---- blocks ------------------------------------------------------------------------------------------------
    def block(self, n):
        """Contiguous block [lo, hi) of n items owned by this rank (equal blocks, the last ones may be short)."""
        per = (n + self.world - 1) // self.world
        lo = min(n, self.rank * per)
        return lo, min(n, lo + per), per

    def gather_blocks(self, local, n):
        """All ranks' blocks of `local` (rows of a 2-D float array, block(n) of n rows each) stacked into [n, ...]."""
        lo, hi, per = self.block(n)
        local = np.ascontiguousarray(local)
        cols = int(np.prod(local.shape[1:], dtype=np.int64)) if local.ndim > 1 else 1
        mine = torch.from_numpy(local.reshape(hi - lo, cols))
        send = torch.zeros([per, cols], dtype=mine.dtype, device=self.device)
        if hi > lo:
            send[: hi - lo] = mine.to(self.device)
        recv = torch.empty([self.world * per, cols], dtype=send.dtype, device=self.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        return recv[:n].cpu().numpy().reshape((n,) + tuple(local.shape[1:]))

    def barrier(self):
        dist.barrier(group=self.group)

    def all_ok(self, ok):
        """True when every rank reports success; a barrier that carries one bit."""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    # ---- sharded roll-up ---------------------------------------------------------------------------------------
    def attach(self, engine, capacity):
        """Shard the engine's symmetric roll-ups over the group: accumulators in a tensor this object owns, summed by
        one integer all-reduce per roll-up on torch's current stream (which the engine is made to launch on)."""
        if self.world == 1 or not hasattr(engine, "set_shard"):
            return False
        dev = torch.device("cuda", engine.device)
        count = 2 * (int(capacity) + 64) + 2
        self._acc = torch.zeros([count], dtype=torch.int64, device=dev)
        engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)

        def allreduce(n, stream):
            # on the stream the library launches on (the one it hands to the hook), whatever torch's current stream is at
            # the moment of the call: the collective then sits between the symmetric kernel and the Euler finisher
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)) if self.backend == "nccl" else _null():
                dist.all_reduce(self._acc[:n], op=dist.ReduceOp.SUM, group=self.group)
        engine.set_shard(self.rank, self.world, allreduce, self._acc.data_ptr(), count * 8, self.min_wake)
        return True

    def detach(self, engine):
        if self._acc is not None:
            try:
                engine.synchronize()
            except Exception:           # (a failed launch must not keep the engine sharded, nor hide the error that led here)
                pass
            engine.set_shard(0, 1)
            self._acc = None


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
