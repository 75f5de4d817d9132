"""GPU tier: the one-process form of a multi-GPU simulation (ludvm_amd/multi.py) on a ONE-GPU box.
RCCL refuses two ranks on one device, so the two halves are exercised apart:
  * the real communicator of the one-process form -- Engine.comm_init_all = ludvm_comm_init_all = ncclCommInitAll -- with ONE
    device, its collectives forced (LUDVM_COMM_FORCE=1): the sharded roll-up's all-reduce is issued on RCCL every step;
  * the threads, the replicas and the sharded roll-up with THREE engines on the one card, joined by a test-only in-process
    all-reduce through ludvm_set_shard's hook (each rank thread meets the others at a barrier inside its C call).
Either way the results must equal the single-engine run BIT FOR BIT (integer sums commute).  With two or more GPUs visible the
real thing runs too: LUDVM(..., devices=2)."""
import threading

import numpy as np
import pytest

from conftest import CONFIG1
from test_multi_threads import MemGroup

pytestmark = pytest.mark.gpu

KW = dict(CONFIG1, tf=8)
SAME = ("Cl", "Cd", "Cm", "LESP", "LEV_shed")


def _engine(dev=0):
    from ludvm_amd import Engine
    e = Engine(dev)
    e.set_symmetric(8)          # symmetric (and overlapped) roll-up steps from 8 vortices on: the README-size case shards
    return e


def _same(a, b):
    return all(np.array_equal(getattr(a, n), getattr(b, n)) for n in SAME) and \
        np.array_equal(a.path["TEV"][a.nt - 1], b.path["TEV"][b.nt - 1]) and np.array_equal(a.circulation["TEV"], b.circulation["TEV"])


class HookGroup(MemGroup):
    """MemGroup + the sharded roll-up: accumulators in a torch tensor per rank, summed across the rank THREADS of this process
    inside ludvm_set_shard's hook (test-only stand-in for the ncclAllReduce of several devices)."""

    def attach(self, engine, capacity):
        import torch
        dev = torch.device("cuda", engine.device)
        count = 2 * (int(capacity) + 64) + 2
        self._acc = torch.zeros([count], dtype=torch.int64, device=dev)
        self._stream = torch.cuda.Stream(dev)
        engine.set_stream(self._stream.cuda_stream)
        sh, rank, world = self.shared, self.rank, self.world

        def allreduce(n, stream):
            s = torch.cuda.ExternalStream(stream, device=dev)
            s.synchronize()                                  # this rank's tile block is summed
            sh["acc"][rank] = self._acc
            sh["barrier"].wait(30)
            with torch.cuda.stream(s):
                total = sh["acc"][0][:n].clone()
                for q in range(1, world):
                    total += sh["acc"][q][:n]
            s.synchronize()
            sh["barrier"].wait(30)                           # everybody has read everybody
            with torch.cuda.stream(s):
                self._acc[:n].copy_(total)
            sh["reduces"][rank] += 1
        engine.set_shard(rank, world, allreduce, self._acc.data_ptr(), count * 8, self.min_wake)
        return True

    def detach(self, engine):
        engine.synchronize()
        engine.set_shard(0, 1)
        engine.set_stream(None)


def hook_join(engines, min_targets, min_wake, min_pairs):
    world = len(engines)
    shared = {"slots": [None] * world, "barrier": threading.Barrier(world), "gathers": [0] * world, "acc": [None] * world,
              "reduces": [0] * world}
    return [HookGroup(shared, r, world, min_targets, min_wake, min_pairs) for r in range(world)]


@pytest.mark.parametrize("march,prec", [(True, "f32"), (False, "f32"), (True, "f32x2")])
def test_three_replica_threads_on_one_card_equal_the_single_engine_run(march, prec):
    from ludvm_amd import LUDVM
    from ludvm_amd.multi import MultiDeviceLUDVM
    kw = dict(KW, verbose=False, precision=prec, history="sparse", march=march)
    one = LUDVM(**kw, engine=_engine())
    multi = MultiDeviceLUDVM((), kw, [0, 1, 2], engine_factory=lambda d: _engine(0), join=hook_join, min_targets=1000, min_wake=64,
                             min_pairs=0)
    try:
        reduces = multi._groups[0].shared["reduces"]
        assert reduces[0] > 100 and len(set(reduces)) == 1          # one all-reduce per sharded step, on every rank
        for rep in multi.replicas():
            assert _same(rep, one)
        assert np.array_equal(multi.Cl, one.Cl)
        if march and prec == "f32":
            args = dict(xmin=-6.0, xmax=1.0, zmin=-1.5, zmax=1.5, dr=0.05, tsteps=[159])
            with pytest.raises(KeyError):
                multi.flowfield(**args)                             # (rows not recorded: the error comes back from the ranks ...)
    finally:
        multi.close()
    assert all(not t.is_alive() for t in multi._team._threads)


def test_flowfield_and_induced_velocity_through_the_front_on_one_card():
    from ludvm_amd import LUDVM
    from ludvm_amd.multi import MultiDeviceLUDVM
    args = dict(xmin=-6.0, xmax=1.0, zmin=-1.5, zmax=1.5, dr=0.05, tsteps=[0, 80, 159])
    kw = dict(KW, verbose=False, precision="f32", history="sparse", snapshot_steps=LUDVM.flowfield_rows_needed(args["tsteps"]))
    one = LUDVM(**kw, engine=_engine())
    one.flowfield(**args)
    with MultiDeviceLUDVM((), kw, [0, 1, 2], engine_factory=lambda d: _engine(0), join=hook_join, min_targets=1000, min_wake=64,
                          min_pairs=0) as multi:
        multi.flowfield(**args)
        for n in ("u_ff", "w_ff", "ome_ff"):                        # row blocks: bit-identical rows (halo rows internal)
            assert np.array_equal(getattr(multi, n), getattr(one, n)), n
        rng = np.random.default_rng(2)
        xw, zw, g = rng.uniform(-3, 0, 5000), rng.uniform(-1, 1, 5000), rng.standard_normal(5000)
        xp, zp = rng.uniform(-3, 0, 9000), rng.uniform(-1, 1, 9000)
        u, w = multi.induced_velocity(g, xw, zw, xp, zp)            # 9000 targets in three blocks
        ur, wr = one.induced_velocity(g, xw, zw, xp, zp)
        scale = max(np.abs(ur).max(), np.abs(wr).max())
        assert max(np.abs(u - ur).max(), np.abs(w - wr).max()) <= 2e-5 * scale


def test_the_one_process_communicator_on_real_rccl_with_one_device(monkeypatch):
    """ludvm_comm_init_all (ncclCommInitAll) over the one device there is, its collectives forced: every sharded roll-up step of
    the run all-reduces its accumulators on RCCL; the bits are the single-engine run's."""
    from ludvm_amd import LUDVM, Engine
    from ludvm_amd._ffi import LudvmHipError
    from ludvm_amd.multi import MultiDeviceLUDVM
    monkeypatch.setenv("LUDVM_COMM_FORCE", "1")
    kw = dict(KW, verbose=False, precision="f32", history="sparse")
    one = LUDVM(**kw, engine=_engine())
    with MultiDeviceLUDVM((), kw, [0], engine_factory=_engine, min_targets=1000, min_wake=64, min_pairs=0) as multi:
        assert multi._engines[0].comm_info() == (0, 1) and multi._groups[0].world == 1
        assert _same(multi.replicas()[0], one)
    assert multi._engines[0].comm_info() == (0, 0)                  # close() left the communicator
    # two contexts on ONE device cannot be ranks of one communicator: refused by the library, with a message
    a, b = Engine(0), Engine(0)
    with pytest.raises(LudvmHipError, match="one context per device"):
        Engine.comm_init_all([a, b], 0)


def test_devices_keyword_with_two_gpus():
    """The real thing: LUDVM(..., devices=2) -- two threads, two devices, ncclCommInitAll, one all-reduce per step over xGMI.
    Runs wherever two GPUs are visible, skips on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL ranks cannot share a card)")
    from ludvm_amd import LUDVM
    from ludvm_amd.multi import MultiDeviceLUDVM
    kw = dict(KW, verbose=False, precision="f32", history="sparse")
    one = LUDVM(**kw, engine=_engine())
    with MultiDeviceLUDVM((), kw, [0, 1], engine_factory=_engine, min_targets=1000, min_wake=64, min_pairs=0) as multi:
        assert all(_same(rep, one) for rep in multi.replicas())
    sim = LUDVM(**dict(CONFIG1, tf=2), verbose=False, devices=2)    # (default thresholds: nothing is large enough to shard)
    assert sim.world == 2 and np.array_equal(sim.Cl, LUDVM(**dict(CONFIG1, tf=2), verbose=False).Cl)
    sim.close()
