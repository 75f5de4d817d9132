"""CPU tier: the host side of the torch-free multi-GPU path (ludvm_amd/comm.py) -- rank discovery from the launcher's
environment, the file rendezvous that carries the communicator identifier from rank 0 to the other ranks, and
LibraryGroup's block partition / gather over an engine stand-in.  (The communicator itself -- ludvm_comm_* on RCCL -- is
exercised in the GPU tier: tests/test_gpu_bench.py::test_library_communicator_one_rank_through_the_c_abi.)"""
import os
import threading

import numpy as np
import pytest

from ludvm_amd import comm


def test_launcher_rank_reads_the_usual_launchers(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK",
              "PMI_RANK", "PMI_SIZE", "SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID"):
        monkeypatch.delenv(k, raising=False)
    assert comm.launcher_rank() == (0, 1, 0)
    monkeypatch.setenv("OMPI_COMM_WORLD_RANK", "3")
    monkeypatch.setenv("OMPI_COMM_WORLD_SIZE", "8")
    monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_RANK", "3")
    assert comm.launcher_rank() == (3, 8, 3)
    monkeypatch.setenv("RANK", "5")            # torchrun's variables win
    monkeypatch.setenv("WORLD_SIZE", "6")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert comm.launcher_rank() == (5, 6, 1)


def test_default_rendezvous_is_per_launch(monkeypatch):
    for k in ("LUDVM_RENDEZVOUS", "LUDVM_LAUNCH_ID", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "OMPI_MCA_ess_base_jobid"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_PORT", "29500")
    a = comm.default_rendezvous()
    monkeypatch.setenv("MASTER_PORT", "29501")
    assert comm.default_rendezvous() != a and str(os.getppid()) in a
    # a launch with a NAME needs no common parent (one ssh / srun step per rank): the name and the port make the file
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")                 # torchrun's default names nothing
    assert comm.launch_name() == "" and str(os.getppid()) in comm.default_rendezvous()
    monkeypatch.setenv("SLURM_JOB_ID", "4711")
    monkeypatch.setenv("SLURM_STEP_ID", "3")
    assert comm.launch_name() == "4711.3" and comm.default_rendezvous().endswith("rdv_4711.3_29501")
    monkeypatch.setenv("LUDVM_LAUNCH_ID", "my run/7")
    assert comm.launch_name() == "my run/7" and comm.default_rendezvous().endswith("rdv_my-run-7_29501")
    assert comm.launch_tag() == b"my run/7|29501"
    monkeypatch.delenv("LUDVM_LAUNCH_ID")
    monkeypatch.delenv("SLURM_JOB_ID")
    d = os.path.dirname(a)                      # a directory of ours that nobody else can write to
    st = os.stat(d)
    assert st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0
    monkeypatch.setenv("LUDVM_RENDEZVOUS", "/somewhere/else")
    assert comm.default_rendezvous() == "/somewhere/else"


def test_identifier_travels_through_the_file(tmp_path):
    path = str(tmp_path / "rdv")
    uid = bytes(range(128))
    got = {}

    def reader(r):
        got[r] = comm.exchange_id(r, None, path, timeout=30)
    threads = [threading.Thread(target=reader, args=(r,)) for r in (1, 2)]
    for t in threads:
        t.start()
    assert comm.exchange_id(0, lambda: uid, path) == uid        # rank 0 publishes (atomically: write + rename)
    for t in threads:
        t.join(30)
    assert got == {1: uid, 2: uid}
    with pytest.raises(TimeoutError):
        comm.exchange_id(1, None, str(tmp_path / "nobody_writes_here"), timeout=0.2)
    assert (os.stat(path).st_mode & 0o077) == 0


def test_rendezvous_does_not_follow_links_nor_trust_stale_files(tmp_path):
    """rank 0 removes what a crashed launch left under the name and never writes through a link; a reader does
    not take an identifier from a link."""
    path = str(tmp_path / "rdv")
    victim = tmp_path / "victim"
    victim.write_bytes(b"precious")
    os.symlink(str(victim), path)                                   # a link where the identifier is about to go
    with pytest.raises(TimeoutError):
        comm.exchange_id(1, None, path, timeout=0.2)                # ... is not read through
    uid = bytes(reversed(range(128)))
    os.symlink(str(victim), f"{path}.{os.getpid()}.tmp")            # ... nor written through
    assert comm.exchange_id(0, lambda: uid, path) == uid
    assert victim.read_bytes() == b"precious" and not os.path.islink(path)
    assert comm.exchange_id(2, None, path, timeout=5) == uid
    # a stale identifier of an earlier launch is replaced, not handed out
    assert comm.exchange_id(0, lambda: bytes(128), path) == bytes(128) and comm.exchange_id(1, None, path, timeout=5) == bytes(128)


def test_a_faster_rank_does_not_take_the_identifier_of_an_earlier_launch(tmp_path, monkeypatch):
    """the file a crashed launch left under the same name (same LUDVM_RENDEZVOUS, or the same parent and port) is
    still there when a non-zero rank of the next launch looks, BEFORE rank 0 has replaced it.  It is older than this launch
    (its mtime predates the launcher's start) -- or carries another launch's tag -- and is waited out, not read."""
    import time
    for k in ("LUDVM_LAUNCH_ID", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "OMPI_MCA_ess_base_jobid"):
        monkeypatch.delenv(k, raising=False)                                   # a launch without a name: the age test decides
    path = str(tmp_path / "rdv")
    stale, fresh = bytes([7]) * 128, bytes(range(128))
    assert comm.exchange_id(0, lambda: stale, path) == stale                   # the earlier launch ...
    le = comm.launch_epoch()
    assert le is not None and le <= time.time() + 1
    past = le - 3600                                                           # ... an hour before our launcher started (ADVICE r5:
    os.utime(path, (past, past))                                               #     derived from the epoch, not from "now")
    with pytest.raises(TimeoutError) as ei:
        comm.exchange_id(1, None, path, timeout=0.3)                           # rank 1 is early: it does NOT join the dead one
    assert "taken for an earlier launch's" in str(ei.value) and "not_before" in str(ei.value)
    got = {}
    t = threading.Thread(target=lambda: got.setdefault(1, comm.exchange_id(1, None, path, timeout=30)))
    t.start()
    __import__("time").sleep(0.2)
    assert comm.exchange_id(0, lambda: fresh, path) == fresh                   # rank 0 arrives and replaces it
    t.join(30)
    assert got == {1: fresh}
    # the same by content: a file of this very minute, but of a launch with another job identifier / port
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-A")
    assert comm.exchange_id(0, lambda: stale, path) == stale
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-B")
    with pytest.raises(TimeoutError) as ei:
        comm.exchange_id(1, None, path, timeout=0.3)
    assert "another launch's tag" in str(ei.value)
    assert comm.exchange_id(0, lambda: fresh, path) == fresh and comm.exchange_id(1, None, path, timeout=5) == fresh


def test_the_launch_epoch_is_a_hint(tmp_path, monkeypatch):
    """ranks started one by one (an ssh / srun step per rank, staggered) have parents YOUNGER than rank 0's file, and a
    container with a virtualised /proc/stat btime reports start times in another clock.  A launch NAME in the tag vouches for
    the file whatever its age; an epoch later than now or later than this process's own start is discarded."""
    import time
    path = str(tmp_path / "rdv")
    uid = bytes(range(128))
    monkeypatch.setenv("LUDVM_LAUNCH_ID", "launch-42")
    assert comm.exchange_id(0, lambda: uid, path) == uid
    old = time.time() - 7200
    os.utime(path, (old, old))
    # (this rank's launcher "started" after rank 0 wrote: with a named launch the file is taken all the same)
    assert comm.exchange_id(1, None, path, timeout=5, not_before=time.time() + 60) == uid
    monkeypatch.delenv("LUDVM_LAUNCH_ID")
    for k in ("TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "OMPI_MCA_ess_base_jobid"):
        monkeypatch.delenv(k, raising=False)
    # a clock that runs ahead (btime virtualised): parent "started" in the future, or after this very process -> no bound
    real = comm._process_start
    monkeypatch.setattr(comm, "_process_start", lambda pid: time.time() + 5000 if pid == os.getppid() else real(pid))
    assert comm.launch_epoch() is None
    monkeypatch.setattr(comm, "_process_start", lambda pid: (real(os.getpid()) or time.time()) + 30 if pid == os.getppid() else real(pid))
    assert comm.launch_epoch() is None
    monkeypatch.setattr(comm, "_process_start", real)
    assert comm.launch_epoch() is not None


def test_ipc_mode_default_for_multi_process_launches(monkeypatch):
    """ONE behaviour on every way into a multi-GPU run -- HSA_ENABLE_IPC_MODE_LEGACY defaults to "0" (dmabuf
    IPC; the pool's image notes, ludvm_amd/comm.py) whenever a launcher announces more than one rank, a value the caller
    exported wins, a lone process is left alone."""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS", comm.IPC_MODE_VAR):
        monkeypatch.delenv(k, raising=False)
    env = {}
    assert comm.prepare_ipc_environment(env) is None and env == {}                 # one process: nothing to share
    assert comm.prepare_ipc_environment(env, force=True) == "0" and env == {comm.IPC_MODE_VAR: "0"}
    env = {comm.IPC_MODE_VAR: "1"}
    assert comm.prepare_ipc_environment(env, force=True) == "1"                    # the caller's choice stands
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert comm.prepare_ipc_environment() == "0" and os.environ[comm.IPC_MODE_VAR] == "0"
    # ... and importing the package under a launcher does it before any HIP call
    import subprocess
    import sys
    from conftest import ROOT
    code = "import os, ludvm_amd; print(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))"
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", comm.IPC_MODE_VAR)}
    run = lambda e: subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=ROOT, timeout=120).stdout.strip()   # noqa: E731
    assert run(base) == "None" and run(dict(base, WORLD_SIZE="2", RANK="1")) == "0"
    assert run(dict(base, WORLD_SIZE="2", RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="1")) == "1"


def test_thresholds_can_be_set_for_a_launch_from_the_environment():
    """`min_wake_suggested` of a bench line applied without editing code: LUDVM_MIN_WAKE / LUDVM_MIN_TARGETS, read at import."""
    import subprocess
    import sys
    from conftest import ROOT
    code = "from ludvm_amd import comm; print(comm.MIN_WAKE, comm.MIN_TARGETS)"
    env = {k: v for k, v in os.environ.items() if k not in ("LUDVM_MIN_WAKE", "LUDVM_MIN_TARGETS")}
    run = lambda e: subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=ROOT, timeout=120).stdout.split()   # noqa: E731
    assert run(env) == ["131072", "65536"]
    assert run(dict(env, LUDVM_MIN_WAKE="65536", LUDVM_MIN_TARGETS="")) == ["65536", "65536"]


class _OneRankEngine:
    """Stand-in for Engine's communicator calls with a group of one."""

    def __init__(self):
        self.calls = []

    def comm_unique_id(self):
        return bytes(128)

    def comm_init(self, rank, world, uid, min_vortices=0):
        self.calls.append(("init", rank, world, len(uid), min_vortices))

    def comm_destroy(self):
        self.calls.append(("destroy",))

    def comm_allgather(self, local):
        return np.ascontiguousarray(local)[None].copy()


def test_library_group_over_an_engine_stand_in(tmp_path):
    eng = _OneRankEngine()
    g = comm.LibraryGroup(eng, rank=0, world=1, rendezvous=str(tmp_path / "rdv"), min_wake=777)
    assert eng.calls == [("init", 0, 1, 128, 777)] and not os.path.exists(str(tmp_path / "rdv"))   # rank 0 removes the file
    rows = np.arange(35.0).reshape(7, 5)
    assert np.array_equal(g.gather_blocks(rows, 7), rows) and g.block(7) == (0, 7, 7)
    g.barrier()
    assert g.attach(eng, 100) is False            # a one-rank group shards nothing
    with pytest.raises(ValueError):
        g.attach(_OneRankEngine(), 100)           # bound to its engine
    g.close()
    assert eng.calls[-1] == ("destroy",)
    # the block partition of a larger group (as ShardGroup: equal blocks, the last ones may be short or empty)
    g3 = comm.LibraryGroup.__new__(comm.LibraryGroup)
    g3.world = 3
    covered = []
    for r in range(3):
        g3.rank = r
        lo, hi, per = g3.block(8)
        assert per == 3
        covered += list(range(lo, hi))
    assert covered == list(range(8))
