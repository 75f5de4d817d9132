"""TEST-ONLY: the shard-step arithmetic with the contract of ludvm_amd.sharded.HipShardKernel, on CPU tensors, by the
oracle (float64 NumPy inside).  What the gloo tests put under ShardedWake to cover the partition, the one collective per
step and the padding without a GPU; tests/bench_cpu_rig.py puts it under bench.py's N > 1 path for the same reason."""
import numpy as np
import torch

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
