"""CPU tier: the one-process form of a multi-GPU simulation (ludvm_amd/multi.py) -- `LUDVM(..., devices=[...])`,
one host thread and one replica per device -- over tests/fake_engine.py and an in-memory group: the thread team, the front
object (replica 0's attributes, collective method calls), the sharding of flowfield rows and induced_velocity targets over the
threads, errors and teardown.  (The real communicator -- ludvm_comm_init_all = ncclCommInitAll -- and the sharded roll-up over
several engines on one card: tests/test_gpu_multi.py.)"""
import threading

import numpy as np
import pytest

from conftest import CONFIG1
from fake_engine import FakeEngine
from ludvm_amd import LUDVM
from ludvm_amd.multi import MultiDeviceLUDVM, ThreadTeam, normalise_devices


class MemGroup:
    """What LUDVM uses of a group (ShardGroup / LibraryGroup), between the threads of one process: an all-gather through a
    shared list and a barrier."""

    def __init__(self, shared, rank, world, min_targets, min_wake, min_pairs):
        self.shared, self.rank, self.world = shared, rank, world
        self.min_targets, self.min_wake, self.min_pairs = min_targets, min_wake, min_pairs
        self.closed = False

    def block(self, n):
        per = (n + self.world - 1) // self.world
        lo = min(n, self.rank * per)
        return lo, min(n, lo + per), per

    def _allgather(self, value):
        self.shared["slots"][self.rank] = value
        self.shared["barrier"].wait(30)
        out = list(self.shared["slots"])
        self.shared["barrier"].wait(30)
        return out

    def gather_blocks(self, local, n):
        self.shared["gathers"][self.rank] += 1
        return np.concatenate([np.asarray(b) for b in self._allgather(np.ascontiguousarray(local))])[:n]

    def barrier(self):
        self._allgather(None)

    def all_ok(self, ok):
        return all(self._allgather(bool(ok)))

    def attach(self, engine, capacity):
        return False                       # (the fake engine has no sharded roll-up: every replica runs the whole loop)

    def detach(self, engine):
        pass

    def close(self):
        self.closed = True


def mem_join(engines, min_targets, min_wake, min_pairs):
    world = len(engines)
    shared = {"slots": [None] * world, "barrier": threading.Barrier(world), "gathers": [0] * world}
    return [MemGroup(shared, r, world, min_targets, min_wake, min_pairs) for r in range(world)]


def test_devices_are_normalised():
    assert normalise_devices(3) == [0, 1, 2] and normalise_devices([2, 0]) == [2, 0] and normalise_devices(np.int64(1)) == [0]
    with pytest.raises(ValueError):
        normalise_devices([1, 1])
    with pytest.raises(ValueError):
        normalise_devices([])


def test_thread_team_runs_ranks_together_and_hands_errors_over():
    team = ThreadTeam(4)
    b = threading.Barrier(4)

    def together(r):
        b.wait(10)                          # (would time out if the ranks ran one after another)
        return r * r, threading.current_thread().name
    out = team.run(together)
    assert [o[0] for o in out] == [0, 1, 4, 9] and len({o[1] for o in out}) == 4
    assert team.run(lambda r: r + 1) == [1, 2, 3, 4]            # the same threads serve the next call

    def fails(r):
        if r in (1, 3):
            raise ValueError(f"rank {r}")
        return r
    with pytest.raises(ValueError, match="rank 1"):             # the lowest failing rank's exception
        team.run(fails)
    with pytest.raises(RuntimeError, match="failed earlier"):   # a rank that failed inside a collective leaves its peers waiting:
        team.run(lambda r: r)                                   # the team is not used again
    team.stop()


@pytest.mark.parametrize("world", [2, 3])
def test_one_process_several_replicas_behind_the_reference_surface(world):
    kw = dict(CONFIG1, tf=3)
    one = LUDVM(**kw, verbose=False, engine=FakeEngine(), precision="f64")
    multi = MultiDeviceLUDVM((), dict(kw, verbose=False, precision="f64"), list(range(world)), engine_factory=lambda d: FakeEngine(),
                             join=mem_join, min_targets=50, min_pairs=0)
    try:
        assert multi.world == world and not isinstance(multi, LUDVM)
        # attributes are replica 0's; every replica holds the same bits
        assert np.array_equal(multi.Cl, one.Cl) and multi.nt == one.nt and multi.itev == one.itev
        for rep in multi.replicas():
            assert np.array_equal(rep.Cl, one.Cl) and np.array_equal(rep.path["TEV"][-1], one.path["TEV"][-1])
        assert multi.replicas()[0]._shard.rank == 0 and multi.replicas()[-1]._shard.rank == world - 1
        # collective methods: flow-field rows in blocks over the threads, gathered on every replica
        args = dict(xmin=-4.0, xmax=1.75, zmin=-1.0, zmax=1.25, dr=0.25, tsteps=[0, 30, 59])
        assert multi.flowfield(**args) is None
        one.flowfield(**args)
        gathers = multi._groups[0].shared["gathers"]
        assert gathers[0] > 0 and len(set(gathers)) == 1                     # every rank took part in every gather
        for name in ("u_ff", "w_ff", "ome_ff"):
            a, b = getattr(multi, name), getattr(one, name)
            assert a.shape == b.shape == (3, 23, 9) and np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(b).max()), name
        # induced_velocity: targets in blocks above min_targets, the calling replica alone below; replica 0's result comes back
        rng = np.random.default_rng(3)
        xw, zw, g = rng.uniform(-3, 0, 200), rng.uniform(-1, 1, 200), rng.standard_normal(200)
        for nt in (601, 37):
            before = list(gathers)
            xp, zp = rng.uniform(-3, 0, nt), rng.uniform(-1, 1, nt)
            u, w = multi.induced_velocity(g, xw, zw, xp, zp)
            ur, wr = one.induced_velocity(g, xw, zw, xp, zp)
            assert np.array_equal(u, ur) and np.array_equal(w, wr)
            assert (gathers[0] - before[0]) == (1 if nt == 601 else 0)
        W = multi.airfoil_downwash(g, xw, zw, 5)
        assert np.array_equal(W, one.airfoil_downwash(g, xw, zw, 5))
        # a parameter the caller sets reaches every replica
        multi.verbose = False
        multi.LESPcrit = 0.15
        assert all(rep.LESPcrit == 0.15 for rep in multi.replicas())
    finally:
        multi.close()
    assert all(g.closed for g in multi._groups) and all(not t.is_alive() for t in multi._team._threads)
    # after close(): the results stay readable, and replica 0 serves further calls alone
    assert np.array_equal(multi.Cl, one.Cl)
    u, w = multi.induced_velocity(g, xw, zw, xp, zp)
    assert np.array_equal(u, ur)
    multi.close()                                                            # idempotent


def test_positional_arguments_as_the_reference_caller_writes_them():
    """LUDVM.py:231-236: the reference's signature can be filled positionally, `verbose` included (position 19)."""
    c = CONFIG1
    args = (c["t0"], 1.0, c["dt"], c["chord"], c["rho"], c["Uinf"], c["Npoints"], c["Ncoeffs"], c["LESPcrit"], c["Naca"],
            None, 1, 2, 0, 10, 0.2 * np.pi, 90, 1, False)
    with MultiDeviceLUDVM(args, dict(precision="f64"), [0, 1], engine_factory=lambda d: FakeEngine(), join=mem_join) as multi:
        one = LUDVM(*args, engine=FakeEngine(), precision="f64")
        assert np.array_equal(multi.Cl, one.Cl) and multi.verbose is False
    with pytest.raises(ValueError, match="do not pass engine"):
        MultiDeviceLUDVM((), dict(CONFIG1, engine=FakeEngine()), [0, 1], engine_factory=lambda d: FakeEngine(), join=mem_join)


def test_a_failure_on_one_replica_reaches_the_caller_and_ends_the_threads():
    made = []

    def factory(d):
        if d == 1:
            raise OSError("no such device [on purpose]")
        made.append(d)
        return FakeEngine()
    n0 = threading.active_count()
    with pytest.raises(OSError, match="on purpose"):
        MultiDeviceLUDVM((), dict(CONFIG1, tf=1, verbose=False), [0, 1, 2], engine_factory=factory, join=mem_join)
    assert sorted(made) == [0, 2]
    for _ in range(50):
        if threading.active_count() <= n0:
            break
        threading.Event().wait(0.1)
    assert threading.active_count() <= n0                                    # the rank threads are gone


def test_devices_keyword_of_the_class():
    """`LUDVM(..., devices=...)`: one device is an ordinary run on it; several go to the one-process front, which needs the
    HIP library and one GPU per replica -- on this GPU-less tier it fails as every engine does, loudly, with no thread left."""
    import torch
    one = LUDVM(**dict(CONFIG1, tf=1), verbose=False, engine=FakeEngine(), precision="f64", devices=[0])
    assert isinstance(one, LUDVM)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    n0 = threading.active_count()
    with pytest.raises(Exception) as ei:
        LUDVM(**dict(CONFIG1, tf=1), verbose=False, devices=2)
    assert "no CPU fallback" in str(ei.value) or "hip" in str(ei.value).lower()
    for _ in range(50):
        if threading.active_count() <= n0:
            break
        threading.Event().wait(0.1)
    assert threading.active_count() <= n0


def test_checkpoints_and_resume_through_the_replicas(tmp_path):
    """A multi-replica run writes its checkpoints once (rank 0; the others wait at the barrier that carries the success bit) and
    `LUDVM.resume(path, devices=...)` resumes every replica from the file: the continuation equals the uninterrupted
    single-engine run bit for bit (here through the builder seam: the fake engines stand in for the devices)."""
    import os
    ck = str(tmp_path / "run.npz")
    kw = dict(CONFIG1, tf=10, verbose=False, history="sparse", snapshot_steps=[50, 120])
    one = LUDVM(**kw, engine=FakeEngine())
    with MultiDeviceLUDVM((), dict(kw, checkpoint_every=70, checkpoint_path=ck), [0, 1, 2], engine_factory=lambda d: FakeEngine(),
                          join=mem_join) as multi:
        assert np.array_equal(multi.Cl, one.Cl) and os.path.exists(ck)
        assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp") or ".tmp." in f]      # one writer, nothing left over
    with MultiDeviceLUDVM((), {}, [0, 1], engine_factory=lambda d: FakeEngine(), join=mem_join,
                          builder=lambda r, eng, grp: LUDVM.resume(ck, engine=eng, verbose=False, distributed=grp)) as res:
        for rep in res.replicas():
            for name in ("Cl", "Cd", "Cm", "LESP", "LEV_shed"):
                assert np.array_equal(getattr(rep, name), getattr(one, name)), name
            assert np.array_equal(np.asarray(rep.path["TEV"][120]), np.asarray(one.path["TEV"][120]))
    with pytest.raises(ValueError, match="do not pass engine"):
        LUDVM.resume(ck, engine=FakeEngine(), devices=[0, 1])
    assert isinstance(LUDVM.resume(ck, engine=FakeEngine(), verbose=False, devices=[0]), LUDVM)   # one device: an ordinary resume
