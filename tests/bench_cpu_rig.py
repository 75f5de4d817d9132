#!/usr/bin/env python3
"""TEST-ONLY: bench.py's control flow on a machine with no GPU.

    python tests/bench_cpu_rig.py --gpus 8 --vortices 17000 ...        (same arguments as bench.py)

runs bench.main() with a stand-in "rig": CPU tensors, gloo, and the pair arithmetic of a shard step by the oracle
(tests/oracle_shard_kernel.py).  Everything else is bench.py's own code: the self-launch of `--gpus N` without a launcher
(the ranks it starts run THIS script, sys.argv[0]), the process group, ShardedWake, the timed regions and their agreement
between the ranks, the budget, both step variants, the result checks, the collective micro-sweep, the deadline, the one JSON
line.  The hangs some tests ask for (LUDVM_BENCH_TEST_HANG*) are injected by this rig (tests/bench_hang_hooks.py), not by bench.py.  The numbers it prints are meaningless as measurements (config.device says so); bench.py itself has no such path and
exits 2 without a GPU.  Used by tests/test_bench_cpu_rehearsal.py to rehearse 2, 4 and 8 ranks.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


class StubEngine:
    """The few engine calls bench.py's config-4 path makes besides the shard kernel; no communicator of its own."""

    def __init__(self):
        self._on, self._ms, self._n = False, 0.0, 0

    def set_tuning(self, *_):
        pass

    def set_symmetric(self, _mode):
        pass

    def device_info(self):
        return {"name": "CPU rehearsal (oracle arithmetic under gloo; NOT a measurement)", "cu_count": 0}

    def kernel_timing(self, enable=True):
        self._on = bool(enable)

    def kernel_time_ms(self, reset=True):
        avg, n = (self._ms / self._n if self._n else 0.0), self._n
        if reset:
            self._ms, self._n = 0.0, 0
        return avg, n

    def comm_unique_id(self):
        raise RuntimeError("the rehearsal rig has no RCCL")

    def _timed(self, fn, *a):
        t0 = time.perf_counter()
        fn(*a)
        if self._on:
            self._ms += (time.perf_counter() - t0) * 1e3
            self._n += 1


class TimedOracleKernel:
    """OracleShardKernel whose pair-sum calls are booked on the stub engine's kernel stopwatch."""

    def __init__(self, eng):
        from oracle_shard_kernel import OracleShardKernel
        self.engine, self._k = eng, OracleShardKernel()
        self.sym_scale, self.advect_from_sums = self._k.sym_scale, self._k.advect_from_sums

    def advect(self, *a):
        self.engine._timed(self._k.advect, *a)

    def sym_accumulate(self, *a):
        self.engine._timed(self._k.sym_accumulate, *a)


from bench_hang_hooks import HangHooks  # noqa: E402


class CpuRig(HangHooks):
    name = "cpu-rehearsal"
    default_backend = "gloo"

    def __init__(self, _local_rank):
        import torch
        self.torch = torch
        torch.set_num_threads(1)
        self.device = torch.device("cpu")

    def new_engine(self):
        return StubEngine()

    def shard_kernel(self, eng):
        return TimedOracleKernel(eng)

    def bind_thread(self):
        pass

    def sync(self):
        pass

    def stopwatch(self):
        return time.perf_counter, lambda a, b: (b - a) * 1e3


if __name__ == "__main__":
    import bench
    bench.main(rig_factory=CpuRig)
