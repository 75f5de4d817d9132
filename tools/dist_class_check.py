#!/usr/bin/env python3
"""One simulation shared by the ranks of a torch.distributed group, checked against the same simulation on one GPU.
Launch with torchrun (one process per GPU; backend nccl = RCCL), or with LUDVM_DIST_BACKEND=gloo to rehearse with
several ranks sharing one card:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 \
        tools/dist_class_check.py

LUDVM_DIST_COLLECTIVES=library: the same check with the library's own RCCL communicator (ludvm_amd/comm.py; the
128-byte identifier travels through a file) -- torch.distributed is then not initialised at all, the launcher only
starts the processes and tells them their rank.

Every rank builds LUDVM(..., distributed=True) with the symmetric threshold and the sharding threshold lowered so that
the README-size case runs sharded roll-ups (tile blocks + one integer all-reduce per step), and compares with a
single-GPU run of its own: loads, circulations and the final wake must agree BIT FOR BIT (integer sums commute), the
sharded flow field too (row blocks), the sharded induced_velocity to fp32 rounding.  Prints DIST_OK <backend> <world>."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine  # noqa: E402
from ludvm_amd.distributed import ShardGroup as _TorchShardGroup  # noqa: E402


def ShardGroup(min_targets, min_wake):        # noqa: N802  (the checks run at small sizes: no lower bound on the pairs of a split call)
    return _TorchShardGroup(min_targets=min_targets, min_wake=min_wake, min_pairs=0)



backend = os.environ.get("LUDVM_DIST_BACKEND", "nccl")
library = os.environ.get("LUDVM_DIST_COLLECTIVES", "torch") == "library"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if library:
    from ludvm_amd.comm import LibraryGroup  # noqa: E402
    backend = "library"

    def ShardGroup(min_targets, min_wake):        # noqa: N802  (same call below; bound to the engine of the moment)
        return LibraryGroup(eng, min_targets=min_targets, min_wake=min_wake, min_pairs=0)
elif backend == "nccl":
    dist.init_process_group(backend="nccl", device_id=dev)
else:
    dist.init_process_group(backend=backend)
assert library or dist.get_world_size() == world
kw = dict(t0=0, tf=8, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")
ok = True
for march in (True, False):
    for prec in ("f32", "f32x2"):
        eng = Engine(local)
        eng.set_symmetric(8)                       # symmetric (and overlapped) steps from 8 vortices on
        sg = ShardGroup(min_targets=1000, min_wake=64)
        # (an engine that owns a communicator stays sharded: the single-GPU reference then runs on an engine of its own)
        eng_one = eng
        if library:
            eng_one = Engine(local)
            eng_one.set_symmetric(8)
        sh = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="sparse", march=march, distributed=sg)
        one = LUDVM(**kw, verbose=False, engine=eng_one, precision=prec, history="sparse", march=march)
        same = all(np.array_equal(getattr(sh, n), getattr(one, n)) for n in ("Cl", "Cd", "Cm", "LESP", "LEV_shed")) and \
            np.array_equal(sh.path["TEV"][sh.nt - 1], one.path["TEV"][one.nt - 1]) and \
            np.array_equal(sh.circulation["TEV"], one.circulation["TEV"])
        print(f"rank {rank} march={march} {prec}: sharded time_loop == single-GPU bit for bit: {same}", flush=True)
        ok &= bool(same)
        if march and prec == "f32":
            args = dict(xmin=-6.0, xmax=1.0, zmin=-1.5, zmax=1.5, dr=0.05, tsteps=[0, 80, 159])
            sg2 = sg if library else ShardGroup(min_targets=1000, min_wake=64)     # (one communicator per engine)
            shf = LUDVM(**kw, verbose=False, engine=eng, precision=prec, history="sparse", distributed=sg2,
                        snapshot_steps=LUDVM.flowfield_rows_needed(args["tsteps"]))
            onef = LUDVM(**kw, verbose=False, engine=eng_one, precision=prec, history="sparse",
                         snapshot_steps=LUDVM.flowfield_rows_needed(args["tsteps"]))
            shf.flowfield(**args)
            onef.flowfield(**args)
            ff = all(np.array_equal(getattr(shf, n), getattr(onef, n)) for n in ("u_ff", "w_ff", "ome_ff"))
            print(f"rank {rank}: sharded flowfield == single-GPU bit for bit: {ff}", flush=True)
            rng = np.random.default_rng(1)
            xw, zw, g = rng.uniform(-5, 0, 3000), rng.uniform(-1, 1, 3000), rng.standard_normal(3000) / 50
            xp, zp = rng.uniform(-5, 0, 5001), rng.uniform(-1, 1, 5001)
            u, w = shf.induced_velocity(g, xw, zw, xp, zp)
            ur, wr = onef.induced_velocity(g, xw, zw, xp, zp)
            iv = max(np.abs(u - ur).max(), np.abs(w - wr).max()) <= 1e-5 * max(np.abs(ur).max(), np.abs(wr).max())
            print(f"rank {rank}: sharded induced_velocity within fp32 rounding: {iv}", flush=True)
            ok &= bool(ff) and bool(iv)
        if library:
            allok = bool(eng.comm_allgather(np.array([1 if ok else 0], np.int8)).min())      # every rank's verdict so far
            ok &= allok
            sg.close()
            eng_one.close()
        eng.close()
# ... and once with the quad variant of the symmetric kernel forced (ludvm_set_sym_tuning(8, -4): four I tiles per workgroup share
# each partner tile; the library's default from 1024 tiles on): the owners then own whole quads of 4 tiles, cut on the device
# from the wake size.  A wake that starts with 9000 free vortices makes it 18+ tiles from the first step.
rng = np.random.default_rng(12)
nfree = 9000
xy = np.stack([rng.uniform(-3.0, -0.5, nfree), rng.uniform(-1.0, 1.0, nfree)], axis=0)
gam = rng.standard_normal(nfree) * 2e-5
kwq = dict(kw, tf=1.5, circulation_freevort=gam, xy_freevort=xy)
from ludvm_amd import _ffi  # noqa: E402
eng = Engine(local, lib_path=_ffi.EXP_LIB_PATH)      # (forcing the variant at this size is a code of the measurement build)
eng.set_symmetric(4096)
eng.set_sym_tuning(8, -4)
sg = ShardGroup(min_targets=1000, min_wake=4096)
eng_one = eng
if library:
    eng_one = Engine(local, lib_path=_ffi.EXP_LIB_PATH)
    eng_one.set_symmetric(4096)
    eng_one.set_sym_tuning(8, -4)
for march in (True, False):
    sh = LUDVM(**kwq, verbose=False, engine=eng, precision="f32", history="sparse", march=march, distributed=sg)
    one = LUDVM(**kwq, verbose=False, engine=eng_one, precision="f32", history="sparse", march=march)
    same = all(np.array_equal(getattr(sh, n), getattr(one, n)) for n in ("Cl", "Cd", "Cm", "LEV_shed")) and \
        np.array_equal(sh.path["FREE"][sh.nt - 1], one.path["FREE"][one.nt - 1])
    print(f"rank {rank} march={march} quad variant: sharded time_loop == single-GPU bit for bit: {same}", flush=True)
    ok &= bool(same)
if library:
    ok &= bool(eng.comm_allgather(np.array([1 if ok else 0], np.int8)).min())
    sg.close()
    eng_one.close()
eng.close()
if library:
    t = torch.tensor([1 if ok else 0])
else:
    t = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
if int(t.item()) != 1:
    sys.exit("sharded run differs from the single-GPU run")
if rank == 0:
    print("DIST_OK", backend, world)
