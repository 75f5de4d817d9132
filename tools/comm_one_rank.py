#!/usr/bin/env python3
"""The library's own RCCL communicator (ludvm_comm_*) with ONE rank, on the real backend, through the C ABI -- no
torch.distributed anywhere.  LUDVM_COMM_FORCE=1 makes the library issue its collectives although a one-rank group would
not need them (they are identities), so every call of the G > 1 paths runs:

  1. LUDVM(..., distributed=LibraryGroup) -- symmetric roll-ups sharded over the communicator, ONE in-library
     ncclAllReduce (int64) per time step, marched and per step -- against the same run without a communicator: bit for bit;
  2. the sharded flow field / induced_velocity gathers (ludvm_comm_allgather_host): bit for bit;
  3. config 4's step (ShardedWake, collectives="library"): ncclAllReduce of the accumulators (symmetric) and ncclAllGather
     of the positions (direct) on device buffers, against the step without collectives: bit for bit.
Prints COMM_OK.  (Two or more ranks: tools/dist_class_check.py with LUDVM_DIST_COLLECTIVES=library on a multi-GPU box.)"""
import os
import sys

os.environ["LUDVM_COMM_FORCE"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402  (device tensors of the config-4 step only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine  # noqa: E402
from ludvm_amd.comm import LibraryGroup  # noqa: E402
from ludvm_amd.sharded import HipShardKernel, ShardedWake  # noqa: E402

kw = dict(t0=0, tf=8, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
ok = True
for march in (True, False):
    for prec in ("f32", "f32x2"):
        eng = Engine(0)
        eng.set_symmetric(8)                       # symmetric (and overlapped) steps from 8 vortices on
        one = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="sparse", march=march)
        grp = LibraryGroup(eng, rank=0, world=1, unique_id=eng.comm_unique_id(), min_targets=1000, min_wake=64)
        assert eng.comm_info() == (0, 1)
        sh = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="sparse", march=march, distributed=grp)
        same = all(np.array_equal(getattr(sh, n), getattr(one, n)) for n in ("Cl", "Cd", "Cm", "LESP", "LEV_shed")) and \
            np.array_equal(sh.path["TEV"][sh.nt - 1], one.path["TEV"][one.nt - 1])
        print(f"march={march} {prec}: time_loop with the in-library all-reduce == without, bit for bit: {same}", flush=True)
        ok &= bool(same)
        if march and prec == "f32":
            rng = np.random.default_rng(1)
            xw, zw, g = rng.uniform(-5, 0, 3000), rng.uniform(-1, 1, 3000), rng.standard_normal(3000) / 50
            xp, zp = rng.uniform(-5, 0, 5001), rng.uniform(-1, 1, 5001)
            # (a one-rank group shards no flow field or induced_velocity call by itself: drive the gather directly)
            blk = grp.gather_blocks(np.stack([xp, zp], axis=1), len(xp))
            ok &= bool(np.array_equal(blk[:, 0], xp) and np.array_equal(blk[:, 1], zp))
            grp.barrier()
        grp.close()
        assert eng.comm_info()[1] == 0
        eng.close()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rng = np.random.default_rng(3)
n = 70000
x, z, g = rng.uniform(-10, 0, n), rng.uniform(-2, 2, n), rng.standard_normal(n) / n
for symmetric in (True, False):
    res = []
    for lib in (False, True):
        eng = Engine(0)
        if lib:
            eng.comm_init(0, 1, eng.comm_unique_id(), min_vortices=1 << 62)
        wake = ShardedWake(x, z, g, 0.065, 5e-2, HipShardKernel(eng), dev, symmetric=symmetric, force_collectives=lib,
                           collectives="library" if lib else "torch")
        for _ in range(3):
            wake.step()
        torch.cuda.synchronize()
        res.append(wake.positions())
        if lib:
            eng.comm_destroy()
        eng.close()
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    print(f"config-4 step, {'symmetric' if symmetric else 'direct'}: in-library collective == none, bit for bit: {same}", flush=True)
    ok &= bool(same)
if not ok:
    sys.exit("the in-library collectives changed a result")
print("COMM_OK")
