"""CPU tier: the multi-GPU shard step (ludvm_amd/sharded.py) under gloo with world_size 2, 3, 4 and 8.
The pair arithmetic is the oracle's (tests only); what is under test is the partition of targets,
the single collective per step (all-gather of positions, or integer all-reduce of the fixed-point sums), the
re-layout of the gathered blocks and padding when N % G != 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ludvm_oracle as O
from oracle_shard_kernel import OracleShardKernel


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


def _worker(rank, world, port, n, steps, out, symmetric):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ludvm_amd.sharded import ShardedWake
        x, z, g = _wake(n)
        wake = ShardedWake(x, z, g, 0.065, 5e-2, OracleShardKernel(), torch.device("cpu"), symmetric=symmetric)
        per = (n + world - 1) // world
        assert wake.n_loc == ((per + 2047) // 2048 * 2048 if symmetric else per) and wake.lo == rank * wake.n_loc   # whole quads of 4 tiles
        assert wake.pairs_per_step == float(n) * n
        wake.collective_timing(True)
        for _ in range(steps):
            wake.step()
        ms, cnt = wake.collective_time_ms()
        assert cnt == steps and ms > 0 and wake.collective_time_ms() == (0.0, 0)      # one collective per step, timed
        xs, zs = wake.positions()
        # the replicas hold the same wake: equal checksums on every rank, through the step's own channel
        sums = wake.gather_checksums()
        assert len(sums) == world and all(len(c) == 4 for c in sums) and wake.ranks_agree()
        bits = xs.view(np.int32).astype(np.int64)
        assert sums[rank][0] == int(bits.sum()) and sums[rank][1] == int((bits * (np.arange(n) % 251 + 1)).sum())
        # ... and a replica that has drifted by one bit in one coordinate is seen by everyone
        if rank == world - 1:
            wake.xs[n // 2] = torch.nextafter(wake.xs[n // 2], torch.tensor(1e9))
        assert not wake.ranks_agree()
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


# world 4 and 8: the machine this is built for has eight GPUs.  (8, 601): more ranks than quads of real vortices -- ranks
# 1-7 own nothing but padding; (8, 5000): blocks of 2048, ranks 3-7 wholly padding, rank 2 partly
@pytest.mark.parametrize("world,n,symmetric", [(2, 600, False), (3, 601, False), (2, 600, True), (3, 601, True),
                                               (2, 2500, True), (4, 2500, True), (8, 5000, True), (8, 5000, False),
                                               (8, 601, True)])
def test_sharded_steps_equal_serial(tmp_path, world, n, symmetric):
    out = str(tmp_path / "pos.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, 3, out, symmetric), nprocs=world, join=True)
    got = np.load(out)
    xr, zr = _serial(n, 3, 0.065, 5e-2)
    np.testing.assert_allclose(got[0], xr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(got[1], zr, rtol=0, atol=2e-6)


@pytest.mark.parametrize("symmetric", [False, True])
def test_single_process_world_of_one(symmetric):
    from ludvm_amd.sharded import ShardedWake
    x, z, g = _wake(300)
    wake = ShardedWake(x, z, g, 0.065, 5e-2, OracleShardKernel(), torch.device("cpu"), symmetric=symmetric)
    wake.step()
    wake.step()
    assert wake.ranks_agree() and len(wake.gather_checksums()) == 1
    a = wake.checksum()
    wake.xs[[3, 7]] = wake.xs[[7, 3]]            # a permutation keeps the plain sums, not the weighted ones
    b = wake.checksum()
    assert a[0] == b[0] and a[1] != b[1] and a[2:] == b[2:]
    wake.xs[[3, 7]] = wake.xs[[7, 3]]
    xr, zr = _serial(300, 2, 0.065, 5e-2)
    xs, zs = wake.positions()
    np.testing.assert_allclose(xs, xr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(zs, zr, rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------------
# flow field: grid rows sharded, halo rows instead of an exchange
# ---------------------------------------------------------------------------------------------
class OracleFlowfieldKernel:
    def flowfield(self, xmin, zmin, dr, nx, nz, xs, zs, gs, v_core, u, w):
        X, Z = np.meshgrid(xmin + np.arange(nx) * dr, zmin + np.arange(nz) * dr, indexing="ij")
        uu, ww = O.induced_velocity(gs.numpy().astype(float), xs.numpy().astype(float), zs.numpy().astype(float),
                                    X.ravel(), Z.ravel(), v_core)
        u.copy_(torch.from_numpy(uu.reshape(nx, nz).astype(np.float32)))
        w.copy_(torch.from_numpy(ww.reshape(nx, nz).astype(np.float32)))

    def vorticity(self, u, w, nx, nz, dr, ome):
        X, Z = np.meshgrid(np.arange(nx) * dr, np.arange(nz) * dr, indexing="ij")
        ome.copy_(torch.from_numpy(O.vorticity(u.numpy()[None].astype(float), w.numpy()[None].astype(float), X, Z)[0]
                                   .astype(np.float32)))


def _ff_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ludvm_amd.sharded import ShardedFlowfield, flowfield_rows
        x, z, g = (torch.from_numpy(a) for a in _wake(200))
        ff = ShardedFlowfield(OracleFlowfieldKernel(), torch.device("cpu"))
        nx, nz = 23, 9
        u, w, ome = ff.compute(-10.0, -2.0, 0.5, nx, nz, x, z, g * 200, 0.065)
        r0, r1 = flowfield_rows(nx, world, rank)
        assert u.shape == (r1 - r0, nz)
        full = [ff.gather(f.contiguous(), nx) for f in (u, w, ome)]
        if rank == 0:
            np.save(out, np.stack([f.numpy() for f in full]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_flowfield_equals_serial(tmp_path, world):
    out = str(tmp_path / "ff.npy")
    mp.spawn(_ff_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    x, z, g = _wake(200)
    k = OracleFlowfieldKernel()
    u, w, ome = (torch.empty([23, 9]) for _ in range(3))
    k.flowfield(-10.0, -2.0, 0.5, 23, 9, torch.from_numpy(x), torch.from_numpy(z), torch.from_numpy(g * 200), 0.065, u, w)
    k.vorticity(u, w, 23, 9, 0.5, ome)
    np.testing.assert_allclose(got[0], u.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got[1], w.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got[2], ome.numpy(), rtol=0, atol=1e-5)
