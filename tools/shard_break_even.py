#!/usr/bin/env python3
"""What a collective may cost before sharding a symmetric roll-up over G GPUs stops paying -- measured on ONE GPU.

For wake sizes around the class-level thresholds (ludvm_amd/comm.py: MIN_WAKE): the symmetric kernel's time over the whole
tile ring (one owner) and over the slowest owner's tile block for G = 2, 4, 8 (ludvm_sym_accumulate_dev_f32, sustained:
`reps` launches back to back), i.e. the pair-kernel time a sharded step SAVES, which is the budget its one all-reduce of
16 N bytes has to fit in.  The collective's own latency over xGMI is not measured here (one GPU per lease).

    python tools/shard_break_even.py [n ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import Engine  # noqa: E402
from ludvm_amd._ffi import SYM_OWNER_ALIGN, SYM_TILE  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [32768, 49152, 65536, 98304, 131072, 196608, 262144, 524288]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng = Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
V_CORE = 0.065


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


for n in sizes:
    rng = np.random.default_rng(n)
    rec = {"n": n, "allreduce_bytes": 16 * n}
    reps = max(20, int(3e5 / (n / 32768.0) ** 2 / 150))           # ~0.3 s of launches per figure
    for G in (1, 2, 4, 8):
        unit = SYM_TILE * SYM_OWNER_ALIGN
        n_loc = ((n + G - 1) // G + unit - 1) // unit * unit
        n_pad = n_loc * G
        pad = n_pad - n
        xs = torch.from_numpy(np.concatenate([rng.uniform(-10, 0, n), np.full(pad, 1e6)]).astype(np.float32)).to(dev)
        zs = torch.from_numpy(np.concatenate([rng.uniform(-2, 2, n), np.full(pad, 1e6)]).astype(np.float32)).to(dev)
        gs = torch.from_numpy(np.concatenate([rng.standard_normal(n) / n, np.zeros(pad)]).astype(np.float32)).to(dev)
        acc = torch.zeros([2 * n_pad + 1], dtype=torch.int64, device=dev)
        scale = torch.zeros([32], dtype=torch.uint8, device=dev)
        eng.sym_scale_dev(gs.data_ptr(), n_pad, V_CORE, scale.data_ptr())
        tiles = n_loc // SYM_TILE
        base = acc.data_ptr()
        owners = []
        for r in range(G):
            owners.append(timed(lambda: eng.sym_accumulate_dev(xs.data_ptr(), zs.data_ptr(), gs.data_ptr(), n_pad, r * tiles, tiles,
                                                               V_CORE, scale.data_ptr(), base, base + 8 * n_pad, base + 16 * n_pad), reps))
        rec[f"G{G}_slowest_owner_us"] = round(max(owners), 1)
        if G > 1:
            rec[f"G{G}_saved_us"] = round(rec["G1_slowest_owner_us"] - max(owners), 1)
    print(json.dumps(rec), flush=True)
