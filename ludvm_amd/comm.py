"""One LUDVM simulation on the GPUs of a node WITHOUT torch.distributed: the collectives run inside libludvm_hip.so on its
own RCCL communicator (ludvm_comm_*).  One process per GPU, started by any launcher that tells each process
its rank (RANK / WORLD_SIZE / LOCAL_RANK as torchrun sets them, or OMPI_COMM_WORLD_*, or SLURM_PROCID / SLURM_NTASKS /
SLURM_LOCALID); every process constructs

    LUDVM(..., device=local_rank, distributed="rccl")            # or distributed=LibraryGroup(engine, ...)

and calls the same methods.  What is sharded and exchanged is what ludvm_amd/distributed.py describes (flow-field rows,
induced_velocity targets, the roll-up's unordered pairs with ONE in-library int64 all-reduce per time step); only the
plumbing differs: the 128-byte communicator identifier travels from rank 0 to the others through a file every rank can
reach (`rendezvous`), the blocks of results through ludvm_comm_allgather_host.
"""
import os
import time

import numpy as np

# From these sizes on the class-level sharding engages (LUDVM(distributed=...)): below them one collective per call / per
# time step costs more than the split saves.  What a sharded roll-up step SAVES is measured (one GPU, the slowest owner's
# tile block against the whole ring: tools/shard_break_even.py, profiles/r04_shard_break_even.txt [MI355X]):
#     wake size      all-reduce     saved per step at G = 2 / 4 / 8
#        32 768        0.5 MB          62 /   95 /   99 us
#        65 536        1   MB         249 /  372 /  405 us
#       131 072        2   MB         957 / 1460 / 1661 us      <- MIN_WAKE: a budget of 1-1.7 ms for a 2 MB all-reduce
#       262 144        4   MB        3843 / 5897 / 6930 us
# The collective's own time over xGMI is NOT yet measured (no multi-GPU box has been available to this build): 131 072 leaves
# it a 10-30x margin over the tens of microseconds RCCL usually needs at that size; 65 536 would leave 3-8x.  Both are
# constructor arguments of ShardGroup / LibraryGroup.  The first `bench.py --gpus N` run on a multi-GPU node measures it:
# its JSON line carries `collective_sweep_us` (the int64 all-reduce at exactly the sizes tabulated above, and the all-gather of
# induced_velocity's blocks) and `min_wake_suggested` (the smallest tabulated wake whose all-reduce costs less than half of
# what the split saves) -- set MIN_WAKE from that key.
# An induced_velocity call is split over the ranks when it has at least MIN_TARGETS targets AND at least MIN_PAIRS pairs: the
# (u, w) blocks come back through one all-gather plus host staging (a few hundred microseconds), which 65 536 targets against
# the 80 bound vortices (5e6 pairs: a microsecond of kernel) can never repay; 2^30 pairs are ~0.2 ms of one GPU's direct kernel.
# LUDVM_MIN_WAKE / LUDVM_MIN_TARGETS / LUDVM_MIN_PAIRS (environment) override the defaults for a whole launch -- e.g. with the
# bench line's `min_wake_suggested` -- without touching code; every rank must see the same values, as with any launcher
# variable.  MIN_WAKE never moves a result bit (the sharded roll-up's integer sums equal the one-GPU sums); a split
# induced_velocity call agrees with the unsplit one within the precision's tolerance, not bit for bit (the partition of a launch
# into partial sums follows its target count).
MIN_TARGETS = int(os.environ.get("LUDVM_MIN_TARGETS", "") or 65536)
MIN_PAIRS = int(os.environ.get("LUDVM_MIN_PAIRS", "") or 2**30)
MIN_WAKE = int(os.environ.get("LUDVM_MIN_WAKE", "") or 131072)


# HSA_ENABLE_IPC_MODE_LEGACY.  Where the default comes from: the description of this build's GPU pool (the ROCm 7.2 image
# notes every build round is given; the variable is exported as 0 in the build container and on the GPU boxes): "the host
# driver only supports dmabuf IPC, and without it RCCL / CUDA-tensor sharing across processes fails with `hipIpcGetMemHandle:
# invalid argument`".  RCCL's intra-node transport maps the peers' buffers through HIP IPC handles, so a process-per-GPU launch
# needs the dmabuf mode there.  No run with two RCCL processes has been possible in this build (one GPU per lease), so the
# claim is the pool's, not a measurement of ours: hence `setdefault` -- a value the user or the launcher exported wins, e.g.
# HSA_ENABLE_IPC_MODE_LEGACY=1 on a host whose driver predates dmabuf IPC (docs/MULTI_GPU_RUNBOOK.md, "If something goes
# wrong").  The HSA runtime reads it when it is initialised, i.e. at the first HIP call of the process: it has to be in the
# environment BEFORE the engine (or torch.cuda) touches the GPU, which is why the package sets it at import time whenever a
# launcher announces more than one rank, bench.py before it imports torch, and its self-launch in the ranks' environment.
IPC_MODE_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"


def prepare_ipc_environment(env=None, force=False):
    """Default HSA_ENABLE_IPC_MODE_LEGACY to "0" (dmabuf IPC) in `env` (default: this process's environment) when a launcher
    announces more than one rank, or always with force=True (an environment being built for ranks).  -> the effective value."""
    env = os.environ if env is None else env
    if force or launcher_rank()[1] > 1:
        env.setdefault(IPC_MODE_VAR, "0")
    return env.get(IPC_MODE_VAR)


def _env_int(*names, default=None):
    for n in names:
        v = os.environ.get(n)
        if v not in (None, ""):
            return int(v)
    return default


def launcher_rank():
    """(rank, world, local_rank) as the process launcher announced them; (0, 1, 0) for a lone process."""
    rank = _env_int("RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID", default=0)
    world = _env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS", default=1)
    local = _env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", default=rank)
    return rank, world, local


def _private_dir():
    """A directory only this user can write to (0700, owned by us), for the rendezvous files: under a shared /tmp another
    local user could otherwise pre-create the file, or plant a symlink where rank 0 is about to write."""
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ludvm_rdv_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise PermissionError(f"{d} exists but is not a directory of ours closed to other users; set LUDVM_RENDEZVOUS")
    return d


def default_rendezvous():
    """A file name all processes of one launch agree on and later launches do not reuse: the launcher's job identifier
    where there is one, else the parent process (the launcher itself) and the rendezvous port -- inside a per-user 0700
    directory."""
    if os.environ.get("LUDVM_RENDEZVOUS"):
        return os.environ["LUDVM_RENDEZVOUS"]
    job = launch_name()
    # (ranks with a launch name need no common parent: one ssh / srun step per rank still agrees on the file)
    tag = f"{job}_{os.environ.get('MASTER_PORT', '0')}" if job else f"_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
    return os.path.join(_private_dir(), "rdv_" + "".join(ch if ch.isalnum() or ch in "._-" else "-" for ch in tag))


def launch_name():
    """A string every rank of ONE launch is told alike and no other launch is: LUDVM_LAUNCH_ID (set it when the ranks are
    started one by one with a shared LUDVM_RENDEZVOUS), torchrun's run identifier unless it is the default "none", the SLURM
    job AND step, Open MPI's job identifier; "" when there is none (a rendezvous port is reused by the next launch: no name)."""
    e = os.environ
    if e.get("LUDVM_LAUNCH_ID"):
        return e["LUDVM_LAUNCH_ID"]
    if e.get("TORCHELASTIC_RUN_ID") and e["TORCHELASTIC_RUN_ID"] != "none":
        return e["TORCHELASTIC_RUN_ID"]
    if e.get("SLURM_JOB_ID"):
        return f"{e['SLURM_JOB_ID']}.{e.get('SLURM_STEP_ID', '')}"
    return e.get("OMPI_MCA_ess_base_jobid") or ""


def _process_start(pid):
    """Wall-clock start of process `pid` (seconds since the epoch, good to a second) from /proc, or None."""
    try:
        with open(f"/proc/{pid}/stat", "rb") as f:
            ticks = int(f.read().rsplit(b")", 1)[1].split()[19])          # field 22, starttime, after "pid (comm)"
        with open("/proc/stat", "rb") as f:
            boot = next(int(line.split()[1]) for line in f if line.startswith(b"btime"))
        return boot + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, StopIteration, IndexError):
        return None


def launch_epoch():
    """A moment no identifier of THIS launch should predate: the start of the process that started us (a common launcher --
    torchrun's agent, orted, slurmstepd -- is older than every rank it starts, rank 0 included), else our own start.  A HINT
    (ADVICE r5): ranks started one by one (an ssh or srun step per rank with a shared LUDVM_RENDEZVOUS) have parents younger
    than rank 0's file, and a container whose /proc/stat btime is virtualised reports start times in another clock -- so a value
    later than now, or later than this process's own start, is discarded (None: no lower bound), and exchange_id accepts a
    file whose non-empty launch tag matches whatever its age."""
    now = time.time()
    own = _process_start(os.getpid())
    t = _process_start(os.getppid())
    if t is None:
        t = own
    if t is None or t > now + 1.0 or (own is not None and t > own + 1.0):
        return None
    return t


def launch_tag():
    """What the launcher tells every rank of one launch alike, as bytes: "<launch name>|<port>" (either may be empty); travels
    behind the identifier so that a reader can tell another launch's file by its content too."""
    return f"{launch_name()}|{os.environ.get('MASTER_PORT', '')}".encode()


def exchange_id(rank, make_id, path, timeout=600.0, not_before=None, tag=None):
    """Rank 0 creates the identifier (make_id() -> bytes) and publishes it atomically at `path`; the others wait for it.
    Rank 0 never writes through a link and first removes whatever a crashed launch left under the name; the readers take
    only a regular file that belongs to this user -- and only one of THIS launch (a non-zero rank that is faster
    than rank 0 could otherwise read the identifier a crashed launch left under the same name, join a dead communicator and
    hang in ncclCommInitRank, which has no timeout): a file whose tag (default: launch_tag()) is another launch's is waited
    out; one whose tag carries this launch's NAME (launch_name(): unique per launch) is taken whatever its age; without a name
    a file last written before `not_before` (default: launch_epoch() -- a hint, see there -- less two seconds for /proc's
    resolution) is waited out as an earlier launch's.  The TimeoutError says what was rejected and why."""
    nofollow = getattr(os, "O_NOFOLLOW", 0)
    tag = launch_tag() if tag is None else bytes(tag)
    if rank == 0:
        uid = make_id()
        try:
            os.unlink(path)                    # a stale identifier would send the other ranks into a dead communicator
        except FileNotFoundError:
            pass
        tmp = f"{path}.{os.getpid()}.tmp"
        try:
            os.unlink(tmp)
        except FileNotFoundError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | nofollow, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(bytes(uid)[:128].ljust(128, b"\0") + tag)
        os.replace(tmp, path)
        return uid
    import stat
    if not_before is None:
        le = launch_epoch()
        not_before = (le - 2.0) if le is not None else 0.0
    # a tag that names the launch vouches for the file by itself (a port alone does not: the next launch reuses it); the age
    # test is for launches that have no name
    tag_names_launch = bool(tag.split(b"|")[0])
    t0 = time.monotonic()
    rejected = None
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | nofollow)
            with os.fdopen(fd, "rb") as f:
                st = os.fstat(f.fileno())
                if stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid():
                    uid = f.read()
                    if len(uid) >= 128 and uid[128:] == tag:
                        if tag_names_launch or st.st_mtime >= not_before:
                            return uid[:128]
                        rejected = f"a file last written at {st.st_mtime:.1f} was taken for an earlier launch's (not_before {not_before:.1f})"
                    elif len(uid) >= 128:
                        rejected = f"a file carries another launch's tag {uid[128:]!r} (ours: {tag!r})"
        except OSError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"no communicator identifier at {path} after {timeout:.0f} s (is rank 0 running?)"
                               + (f"; {rejected}" if rejected else ""))
        time.sleep(0.01)


class LibraryGroup:
    """The ranks that share one simulation, joined by the engine's own RCCL communicator.  Same surface as
    ludvm_amd.distributed.ShardGroup (what LUDVM uses of it), no torch."""

    backend = "rccl (in-library)"
    _created = 0          # groups this process has created through a default rendezvous: every rank creates them in the same
                          # order, so the count names the file (a rank must not pick up the identifier of the previous group)

    def __init__(self, engine, rank=None, world=None, rendezvous=None, unique_id=None, min_targets=MIN_TARGETS, min_wake=MIN_WAKE,
                 timeout=600.0, min_pairs=MIN_PAIRS):
        r, w, _ = launcher_rank()
        self.engine = engine
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.min_targets, self.min_wake, self.min_pairs = int(min_targets), int(min_wake), int(min_pairs)
        self._path = None
        if unique_id is None:
            if rendezvous is None:
                rendezvous = f"{default_rendezvous()}.{LibraryGroup._created}"
                LibraryGroup._created += 1
            self._path = rendezvous
            unique_id = exchange_id(self.rank, engine.comm_unique_id, self._path, timeout)
        engine.comm_init(self.rank, self.world, unique_id, self.min_wake)      # returns when every rank has joined
        if self._path and self.rank == 0:
            try:
                os.remove(self._path)           # everyone has read it
            except OSError:
                pass

    @classmethod
    def joined(cls, engine, rank, world, min_targets=MIN_TARGETS, min_wake=MIN_WAKE, min_pairs=MIN_PAIRS):
        """The group of an engine that HAS joined its communicator already (Engine.comm_init_all: the one-process form,
        ludvm_amd/multi.py) -- no rendezvous, no identifier."""
        g = cls.__new__(cls)
        g.engine, g.rank, g.world = engine, int(rank), int(world)
        g.min_targets, g.min_wake, g.min_pairs = int(min_targets), int(min_wake), int(min_pairs)
        g._path = None
        return g

    def close(self):
        if self.engine is not None:
            self.engine.comm_destroy()
            self.engine = None

    # ---- blocks (as ShardGroup) ------------------------------------------------------------------------------------
    def block(self, n):
        per = (n + self.world - 1) // self.world
        lo = min(n, self.rank * per)
        return lo, min(n, lo + per), per

    def gather_blocks(self, local, n):
        lo, hi, per = self.block(n)
        local = np.ascontiguousarray(local)
        send = np.zeros((per,) + local.shape[1:], local.dtype)
        send[: hi - lo] = local
        allb = self.engine.comm_allgather(send)                       # [world, per, ...]
        return allb.reshape((self.world * per,) + local.shape[1:])[:n]

    def barrier(self):
        self.engine.comm_allgather(np.zeros(1, np.int8))

    def all_ok(self, ok):
        """True when every rank reports success; a barrier that carries one bit."""
        return bool(np.all(self.engine.comm_allgather(np.array([1 if ok else 0], np.int8)) != 0))

    # ---- sharded roll-up: the communicator already shards the engine ------------------------------------------------------
    def attach(self, engine, capacity):
        if engine is not self.engine:
            raise ValueError("LibraryGroup is bound to the engine it was created with")
        return self.world > 1

    def detach(self, engine):
        """Nothing to undo: the communicator shards the engine for as long as it exists (from min_wake vortices on), i.e.
        until close() -- also for roll-ups issued between two time loops.  (ShardGroup.detach unshards its engine.)"""

