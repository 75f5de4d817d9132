"""CPU tier: the multi-GPU shard step (ludvm_amd/sharded.py) under gloo with world_size 2 and 3.
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


class OracleShardKernel:
    """The shard-step arithmetic with the same contract as HipShardKernel, on CPU tensors (float64
    NumPy inside).  sym_accumulate follows the symmetric kernel's assignment of unordered pairs to
    I-tiles: J = I + d (mod NT), d = 1..(NT-1)/2, the half-way offset of an even ring taken by the
    lower half only, the diagonal tile evaluated ordered."""

    def advect(self, xs, zs, gs, t_first, nt, v_core, dt, x_out, z_out):
        x, z, g = xs.numpy().astype(np.float64), zs.numpy().astype(np.float64), gs.numpy().astype(np.float64)
        sl = slice(t_first, t_first + nt)
        u, w = O.induced_velocity(g, x, z, x[sl], z[sl], v_core)
        x_out.copy_(torch.from_numpy((x[sl] + dt * u).astype(np.float32)))
        z_out.copy_(torch.from_numpy((z[sl] + dt * w).astype(np.float32)))

    def sym_scale(self, gs, v_core, scale):
        """The library's rule: a power of two that keeps sum|Gamma| / (sqrt(2) v_core) under 2^61 (opaque record;
        here: float64 [S, 1/S])."""
        bound = float(np.abs(gs.numpy().astype(np.float64)).sum()) / (np.sqrt(2.0) * v_core)
        k = 61 - (int(np.frexp(bound)[1]) if bound > 0 else 0)
        scale.view(torch.float64)[0] = 2.0 ** k
        scale.view(torch.float64)[1] = 2.0 ** -k

    def sym_accumulate(self, xs, zs, gs, tile_first, tile_count, v_core, scale, acc):
        x, z, g = xs.numpy().astype(np.float64), zs.numpy().astype(np.float64), gs.numpy().astype(np.float64)
        from ludvm_amd._ffi import SYM_TILE
        n, W = len(x), SYM_TILE
        nt = (n + W - 1) // W
        even = nt % 2 == 0 and nt > 1
        dtot = (nt - 1) // 2 + (1 if even else 0)
        S = float(scale.view(torch.float64)[0])
        au, aw = np.zeros(n, np.int64), np.zeros(n, np.int64)

        def fx(v):          # one fp32 partial sum -> fixed point, as the kernel adds it
            return np.trunc(v.astype(np.float32).astype(np.float64) * S).astype(np.int64)

        def block(isl, jsl):
            dx = x[isl, None] - x[None, jsl]
            dz = z[isl, None] - z[None, jsl]
            s = 1.0 / np.sqrt((dx * dx + dz * dz) ** 2 + v_core**4)
            return dx, dz, s

        for I in range(tile_first, tile_first + tile_count):
            isl = slice(I * W, min(n, (I + 1) * W))
            dx, dz, s = block(isl, isl)                     # diagonal tile: ordered, i-side only
            au[isl] += fx((g[None, isl] * dz * s).sum(1))
            aw[isl] += fx((g[None, isl] * dx * s).sum(1))
            for d in range(1, dtot + 1):
                if even and d == dtot and I >= nt // 2:
                    break
                J = (I + d) % nt
                jsl = slice(J * W, min(n, (J + 1) * W))
                dx, dz, s = block(isl, jsl)
                au[isl] += fx((g[None, jsl] * dz * s).sum(1))
                aw[isl] += fx((g[None, jsl] * dx * s).sum(1))
                au[jsl] -= fx((g[isl, None] * dz * s).sum(0))   # j feels the opposite of what i feels
                aw[jsl] -= fx((g[isl, None] * dx * s).sum(0))
        acc[:n] += torch.from_numpy(au)
        acc[n:2 * n] += torch.from_numpy(aw)

    def advect_from_sums(self, acc, scale, xs, zs, t_first, nt, dt, x_out, z_out):
        n = xs.numel()
        assert int(acc[2 * n]) == 0
        inv = float(scale.view(torch.float64)[1])
        sl = slice(t_first, t_first + nt)
        k = 1.0 / (2 * np.pi)
        su = (acc[:n][sl].numpy().astype(np.float64) * inv).astype(np.float32)
        sw = (acc[n:2 * n][sl].numpy().astype(np.float64) * inv).astype(np.float32)
        x_out.copy_(xs[sl] + dt * torch.from_numpy(su * np.float32(k)))
        z_out.copy_(zs[sl] - dt * torch.from_numpy(sw * np.float32(k)))


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


@pytest.mark.parametrize("world,n,symmetric", [(2, 600, False), (3, 601, False), (2, 600, True), (3, 601, True),
                                               (2, 2500, True)])
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


@pytest.mark.parametrize("world", [2, 3])
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
