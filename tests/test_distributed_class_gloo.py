"""CPU tier: `LUDVM(..., distributed=True)` under gloo with world_size 2, 3 and 8 -- the sharding of LUDVM.flowfield
(grid rows) and LUDVM.induced_velocity (targets) behind the reference's method surface, and the gather that leaves the
reference's full arrays on every rank.  The pair arithmetic is the oracle's (tests/fake_engine.py); what is under test
is the partition, the halo rows of the vorticity stencil, ragged blocks and the collectives."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CONFIG1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import FakeEngine
        from ludvm_amd import LUDVM
        from ludvm_amd.distributed import ShardGroup
        kw = dict(CONFIG1, tf=3)
        sg = ShardGroup(min_targets=50, min_pairs=0)
        assert (sg.world, sg.rank, sg.backend) == (world, rank, "gloo")
        sim = LUDVM(**kw, verbose=False, engine=FakeEngine(), precision="f64", distributed=sg)
        one = LUDVM(**kw, verbose=False, engine=FakeEngine(), precision="f64")
        # time loop: the fake engine has no sharded roll-up, every rank ran the whole loop -> identical results
        assert np.array_equal(sim.Cl, one.Cl)
        # flow field: 23 rows over 2, 3 or 8 ranks (ragged; 8 ranks: blocks of 3 rows, the last one 2), 9 columns, three time steps incl. step 0 (free vortices only)
        args = dict(xmin=-4.0, xmax=1.75, zmin=-1.0, zmax=1.25, dr=0.25, tsteps=[0, 30, 59])
        sim.flowfield(**args)
        one.flowfield(**args)
        assert sim.u_ff.shape == one.u_ff.shape == (3, 23, 9)
        for name in ("u_ff", "w_ff", "ome_ff"):
            a, b = getattr(sim, name), getattr(one, name)
            assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(b).max()), name      # float32 transport of the blocks
        # induced_velocity: 601 targets in blocks (above min_targets), 37 below it (stays local)
        rng = np.random.default_rng(3)
        xw, zw, g = rng.uniform(-3, 0, 200), rng.uniform(-1, 1, 200), rng.standard_normal(200)
        for nt in (601, 37):
            xp, zp = rng.uniform(-3, 0, nt), rng.uniform(-1, 1, nt)
            u, w = sim.induced_velocity(g, xw, zw, xp, zp)
            ur, wr = one.induced_velocity(g, xw, zw, xp, zp)
            assert u.shape == (nt,) and np.array_equal(u, ur) and np.array_equal(w, wr)
        # ... and many targets against few sources stay local too: min_pairs (default 2^30) -- the gather is never issued
        few = ShardGroup(min_targets=50)
        assert few.min_pairs == 2**30
        simf = LUDVM(**dict(kw, tf=0.2), verbose=False, engine=FakeEngine(), precision="f64", distributed=few)
        calls = []
        few.gather_blocks = lambda *a, **k: calls.append(a) or (_ for _ in ()).throw(AssertionError("split a call of 1.2e5 pairs"))
        xp, zp = rng.uniform(-3, 0, 601), rng.uniform(-1, 1, 601)
        u, w = simf.induced_velocity(g, xw, zw, xp, zp)
        ur, wr = one.induced_velocity(g, xw, zw, xp, zp)
        assert not calls and np.array_equal(u, ur) and np.array_equal(w, wr)
        if rank == 0:
            np.save(out, sim.u_ff)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_class_level_sharding_equals_one_rank(tmp_path, world):
    out = str(tmp_path / "u.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out).shape == (3, 23, 9)


def test_distributed_needs_a_process_group():
    from fake_engine import FakeEngine
    from ludvm_amd import LUDVM
    with pytest.raises(RuntimeError):
        LUDVM(**dict(CONFIG1, tf=0.2), verbose=False, engine=FakeEngine(), distributed=True)


def _ckpt_worker(rank, world, port, ck):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import FakeEngine
        from ludvm_amd import LUDVM
        from ludvm_amd.distributed import ShardGroup
        kw = dict(CONFIG1, tf=3)
        ref = LUDVM(**kw, verbose=False, engine=FakeEngine(), precision="f64")
        # every rank runs the whole loop and reaches the checkpoint steps together: rank 0 alone writes the file (the ranks
        # would race on the temporary file), the others wait at the barrier behind it
        sim = LUDVM(**kw, verbose=False, engine=FakeEngine(), precision="f64", distributed=ShardGroup(), checkpoint_every=20,
                    checkpoint_path=ck)
        assert np.array_equal(sim.Cl, ref.Cl)
        dist.barrier()
        assert os.path.exists(ck) and not os.path.exists(ck + ".tmp.npz")
        # ... and a shared run resumes shared, from the same file on every rank
        res = LUDVM.resume(ck, engine=FakeEngine(), verbose=False, distributed=ShardGroup())
        assert res._shard is not None and res._shard.world == world
        assert np.array_equal(res.Cl, ref.Cl) and np.array_equal(res.LEV_shed, ref.LEV_shed)
    finally:
        dist.destroy_process_group()


def test_shared_run_writes_its_checkpoint_once_and_resumes_shared(tmp_path):
    mp.spawn(_ckpt_worker, args=(2, _free_port(), str(tmp_path / "ck.npz")), nprocs=2, join=True)


def _ckpt_fail_worker(rank, world, port, ck):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import FakeEngine
        from ludvm_amd import LUDVM
        from ludvm_amd.distributed import ShardGroup
        # the checkpoint path lies in a directory that does not exist: rank 0's write fails -- and EVERY rank raises, at the
        # same step, instead of rank 0 raising alone while the others wait for it in the barrier
        with pytest.raises((OSError, RuntimeError)) as ei:
            LUDVM(**dict(CONFIG1, tf=3), verbose=False, engine=FakeEngine(), precision="f64", distributed=ShardGroup(),
                  checkpoint_every=20, checkpoint_path=ck)
        assert isinstance(ei.value, OSError) == (rank == 0)
        dist.barrier()          # both ranks got here: nobody hangs
    finally:
        dist.destroy_process_group()


def test_a_failed_checkpoint_write_raises_on_every_rank(tmp_path):
    mp.spawn(_ckpt_fail_worker, args=(2, _free_port(), str(tmp_path / "no_such_dir" / "ck.npz")), nprocs=2, join=True)
