"""CPU tier: the multi-GPU shard step (ludvm_amd/sharded.py) under gloo with world_size 2 and 3.
The pair arithmetic is the oracle's (tests only); what is under test is the partition of targets,
the single all-gather, the re-layout of the gathered blocks and padding when N % G != 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ludvm_oracle as O


class OracleShardKernel:
    """advect() with the same contract as HipShardKernel, on CPU tensors."""

    def advect(self, xs, zs, gs, t_first, nt, v_core, dt, x_out, z_out):
        x, z, g = xs.numpy().astype(np.float64), zs.numpy().astype(np.float64), gs.numpy().astype(np.float64)
        sl = slice(t_first, t_first + nt)
        u, w = O.induced_velocity(g, x, z, x[sl], z[sl], v_core)
        x_out.copy_(torch.from_numpy((x[sl] + dt * u).astype(np.float32)))
        z_out.copy_(torch.from_numpy((z[sl] + dt * w).astype(np.float32)))


def _wake(n):
    rng = np.random.default_rng(5)
    return (rng.uniform(-10, 0, n).astype(np.float32), rng.uniform(-2, 2, n).astype(np.float32),
            (rng.standard_normal(n) / n).astype(np.float32))


def _serial(n, steps, v_core, dt):
    x, z, g = _wake(n)
    x, z, g = x.astype(np.float64), z.astype(np.float64), g.astype(np.float64)
    for _ in range(steps):
        u, w = O.induced_velocity(g, x, z, x, z, v_core)
        x = (x + dt * u).astype(np.float32).astype(np.float64)
        z = (z + dt * w).astype(np.float32).astype(np.float64)
    return x, z


def _worker(rank, world, port, n, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ludvm_amd.sharded import ShardedWake
        x, z, g = _wake(n)
        wake = ShardedWake(x, z, g, 0.065, 5e-2, OracleShardKernel(), torch.device("cpu"))
        assert wake.n_loc == (n + world - 1) // world and wake.lo == rank * wake.n_loc
        assert wake.pairs_per_step == float(n) * n
        for _ in range(steps):
            wake.step()
        xs, zs = wake.positions()
        if rank == 0:
            np.save(out, np.stack([xs, zs]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n", [(2, 600), (3, 601)])
def test_sharded_steps_equal_serial(tmp_path, world, n):
    out = str(tmp_path / "pos.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, 3, out), nprocs=world, join=True)
    got = np.load(out)
    xr, zr = _serial(n, 3, 0.065, 5e-2)
    np.testing.assert_allclose(got[0], xr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(got[1], zr, rtol=0, atol=2e-6)


def test_single_process_world_of_one():
    from ludvm_amd.sharded import ShardedWake
    x, z, g = _wake(300)
    wake = ShardedWake(x, z, g, 0.065, 5e-2, OracleShardKernel(), torch.device("cpu"))
    wake.step()
    wake.step()
    xr, zr = _serial(300, 2, 0.065, 5e-2)
    xs, zs = wake.positions()
    np.testing.assert_allclose(xs, xr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(zs, zr, rtol=0, atol=2e-6)
