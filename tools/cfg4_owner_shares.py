#!/usr/bin/env python3
"""BASELINE config 4 (N = 8e6, sharded over G GPUs) rehearsed on ONE GPU: the time of every owner's share of a step,
for G = 1, 2, 4, 8 and both variants, plus the parts every owner repeats.  No collective runs here and nothing is a
scaling measurement -- it shows how evenly the work divides and what the replicated O(N) parts cost, i.e. the compute
side of the N-GPU step (the collective adds one all-reduce of 128 MB or one all-gather of 64 MB per step over xGMI).

symmetric variant (default of bench.py --gpus G): owner r evaluates tile block r of the unordered pairs into the
  full-length fixed-point accumulators (ludvm_sym_accumulate_dev_f32); every owner also zeroes the accumulators and
  Euler-updates all N vortices from the reduced sums (ludvm_advect_from_sums_dev_f32).
direct variant (--symmetric 0): owner r evaluates all N sources on its own N / G targets with the Euler update fused
  (ludvm_advect_dev_f32).

    python tools/cfg4_owner_shares.py [--vortices 8000000] [--owners 1 2 4 8]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import Engine  # noqa: E402
from ludvm_amd._ffi import SYM_OWNER_ALIGN, SYM_TILE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--vortices", type=int, default=8_000_000)
ap.add_argument("--owners", type=int, nargs="+", default=[1, 2, 4, 8])
args = ap.parse_args()
n = args.vortices
rng = np.random.default_rng(20260101)
x = rng.uniform(-10, 0, n).astype(np.float32)
z = rng.uniform(-2, 2, n).astype(np.float32)
g = (rng.standard_normal(n) / n).astype(np.float32)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng = Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
V_CORE, DT = 0.065, 5e-2


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


print(f"config 4 on one GPU, N = {n}: per-owner share of one step [ms] ({eng.device_info()['name']})")
for G in args.owners:
    unit = SYM_TILE * SYM_OWNER_ALIGN
    n_loc = ((n + G - 1) // G + unit - 1) // unit * unit
    n_pad = n_loc * G
    pad = n_pad - n
    xs = torch.from_numpy(np.concatenate([x, np.full(pad, 1e6, np.float32)])).to(dev)
    zs = torch.from_numpy(np.concatenate([z, np.full(pad, 1e6, np.float32)])).to(dev)
    gs = torch.from_numpy(np.concatenate([g, np.zeros(pad, np.float32)])).to(dev)
    acc = torch.zeros([2 * n_pad + 1], dtype=torch.int64, device=dev)
    scale = torch.zeros([32], dtype=torch.uint8, device=dev)
    xo, zo = torch.empty_like(xs), torch.empty_like(zs)
    eng.sym_scale_dev(gs.data_ptr(), n_pad, V_CORE, scale.data_ptr())
    tiles = n_loc // SYM_TILE
    base = acc.data_ptr()
    # warm both kernels
    eng.sym_accumulate_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), n_pad, 0, min(tiles, 64), V_CORE, scale.data_ptr(), base,
                           base + 8 * n_pad, base + 16 * n_pad)
    torch.cuda.synchronize()
    sym = []
    for r in range(G):
        acc.zero_()
        torch.cuda.synchronize()
        sym.append(timed(lambda: eng.sym_accumulate_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), n_pad, r * tiles, tiles, V_CORE,
                                                        scale.data_ptr(), base, base + 8 * n_pad, base + 16 * n_pad)))
    t_zero = timed(lambda: acc.zero_())
    t_euler = timed(lambda: eng.advect_from_sums_dev(base, base + 8 * n_pad, scale.data_ptr(), base + 16 * n_pad, xs.data_ptr(),
                                                     zs.data_ptr(), 0, n_pad, DT, xo.data_ptr(), zo.data_ptr()))
    eng.set_symmetric(0)
    direct = []
    send = torch.empty([2, n_loc], dtype=torch.float32, device=dev)
    for r in range(G):
        direct.append(timed(lambda: eng.advect_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), n_pad, r * n_loc, n_loc, V_CORE, DT,
                                                   send[0].data_ptr(), send[1].data_ptr())))
    eng.set_symmetric(1)
    s, d = np.array(sym), np.array(direct)
    pairs = float(n) * n
    print(f"G = {G}: N/G = {n_loc}, {tiles} tiles per owner")
    print("  symmetric  shares " + " ".join(f"{v:8.1f}" for v in s) + f"   max {s.max():.1f}  mean {s.mean():.1f}  max/mean {s.max() / s.mean():.4f}")
    print(f"             replicated per owner: zero accumulators {t_zero:.2f}  Euler update of all N {t_euler:.2f}"
          f"   -> step (compute) {s.max() + t_zero + t_euler:.1f} ms = {pairs / ((s.max() + t_zero + t_euler) * 1e-3):.3e} pairs/s,"
          f" {s.sum():.1f} ms of pair work in all")
    print("  direct     shares " + " ".join(f"{v:8.1f}" for v in d) + f"   max {d.max():.1f}  mean {d.mean():.1f}  max/mean {d.max() / d.mean():.4f}"
          f"   -> step (compute) {d.max():.1f} ms = {pairs / (d.max() * 1e-3):.3e} pairs/s, {d.sum():.1f} ms in all")
    del xs, zs, gs, acc, xo, zo, send
