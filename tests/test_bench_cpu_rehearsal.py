"""CPU tier: bench.py's N > 1 control flow with 2 (the deadline test) and 8 ranks, on a machine with no GPU.

`python bench.py --gpus N` without a launcher starts its own ranks; tests/bench_cpu_rig.py runs
bench.main() with the oracle's arithmetic under gloo in place of the HIP engine, so everything else -- the self-launch, the
process group, the tile blocks of ShardedWake (at 8 ranks: blocks that are partly or wholly padding), the agreement of the
ranks on step counts, both step variants, the result checks, the collective micro-sweep, the deadline -- is bench.py's own
code.  bench.py itself has no CPU path: the last tests show it refusing to run here."""
import json
import os
import subprocess
import sys
import time

import pytest

from bench_checks import assert_self_checking_config4, assert_sweep
from conftest import ROOT

RIG = os.path.join(ROOT, "tests", "bench_cpu_rig.py")


def _env(**extra):
    env = dict(os.environ, **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LUDVM_BENCH_BACKEND"):
        env.pop(k, None)
    return env


def _run(script, *args, env=None, timeout=600):
    p = subprocess.run([sys.executable, script, *args], capture_output=True, text=True, env=env or _env(), timeout=timeout)
    return p, [l for l in p.stdout.splitlines() if l.strip()]


@pytest.mark.parametrize("ranks,n", [(8, 17000)])          # (four self-launched ranks: tests/test_gpu_bench.py, on the card)
def test_self_launched_ranks_rehearse_config4(ranks, n):
    """Plain `<bench> --gpus N`: N ranks come up, one line comes back.  17 000 vortices on 8 ranks: blocks of 4096 (whole
    quads of 512-vortex tiles), ranks 5-7 own nothing but padding; the bits are those of any other world size."""
    p, lines = _run(RIG, "--gpus", str(ranks), "--vortices", str(n), "--steps", "1", "--warmup", "0", "--repeats", "0")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["ranks"] == ranks and d["scaling"] == "strong" and "incomplete" not in d
    assert "bench.py itself" in d["config"]["launched_by"] and "torch.distributed.run" in p.stderr
    assert "config 4" in d["config"]["workload"] and d["config"]["collective_backend"] == "gloo"
    # ONE behaviour on every way in: the ranks' environment carries the IPC mode, the line echoes it
    assert d["config"]["hsa_enable_ipc_mode_legacy"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert abs(d["value"] - float(n) ** 2 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["repeat_values"] == [] and "cpu_baseline" not in d and "config4_one_gpu" not in d
    assert_self_checking_config4(d, ranks, min_value=1e6)
    assert_sweep(d, ranks, ["torch"])
    # the same wake on any number of ranks: the same bits (integer sums commute) -- what two ranks print for the same arguments
    p2, lines2 = _run(RIG, "--gpus", "2", "--vortices", str(n), "--steps", "1", "--warmup", "0", "--repeats", "0", "--other-variant", "0",
                      "--sweep", "0")
    assert p2.returncode == 0 and len(lines2) == 1, p2.stderr[-3000:]
    d2 = json.loads(lines2[0])
    assert d2["config"]["ranks"] == 2 and "collective_sweep_us" not in d2 and "direct_variant" not in d2
    assert d2["result_check"]["symmetric"]["checksum"] == d["result_check"]["symmetric"]["checksum"]


def test_self_launch_relays_the_deadline_line():
    """A phase that never ends (on purpose, after the reported region), two self-launched ranks: at --deadline-s rank 0 prints
    the line as it stands, every rank exits 0, the launcher exits 0 and the parent passes the one line on."""
    t0 = time.time()
    p, lines = _run(RIG, "--gpus", "2", "--vortices", "16384", "--steps", "1", "--warmup", "0", "--deadline-s", "25",
                    env=_env(LUDVM_BENCH_TEST_HANG="1"), timeout=200)
    assert p.returncode == 0, p.stderr[-3000:]
    assert 20 < time.time() - t0 < 90
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "deadline" in d["incomplete"] and "hung on purpose" in d["incomplete"] and d["n_gpus"] == 2 and d["value"] > 0
    assert "deadline of 25 s reached" in p.stderr


def test_deadline_before_anything_was_measured_is_an_error():
    """Nothing measured when the deadline comes: no line, exit status 3 (a harness must not take that for a run)."""
    env = _env(LUDVM_BENCH_TEST_HANG="1")
    p = subprocess.run([sys.executable, "-c", "import sys, time; sys.argv = ['x']; import bench; r = bench.Reporter(0, 1.0, "
                        "time.perf_counter()); time.sleep(30)"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=60)
    assert p.returncode == 3 and p.stdout == "" and "deadline of 1 s reached during: start-up" in p.stderr


def test_the_measurement_program_reads_no_test_variable():
    """bench.py's N > 1 path is frozen and carries no test hook -- the hangs above are injected by the rigs
    (tests/bench_hang_hooks.py through HipRig.at_milestone / at_comm_join, which do nothing on the machine)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "LUDVM_BENCH_TEST" not in src
    import bench
    assert bench.HipRig.at_milestone(None, None, "x") is None and bench.HipRig.at_comm_join(None, "before", 1.0) is None


def test_launcher_world_that_contradicts_gpus_is_refused():
    p, lines = _run(RIG, "--gpus", "4", env=dict(_env(), WORLD_SIZE="2", RANK="0"))
    assert p.returncode == 2 and lines == [] and "WORLD_SIZE is 2" in p.stderr
    # ... while a bare WORLD_SIZE=1 that some shell exports is no launcher: the ranks are started all the same
    p, lines = _run(RIG, "--gpus", "2", "--vortices", "16384", "--steps", "1", "--warmup", "0", "--repeats", "0", "--other-variant", "0",
                    "--sweep", "0", "--check", "0", env=dict(_env(), WORLD_SIZE="1"))
    assert p.returncode == 0 and len(lines) == 1 and json.loads(lines[0])["config"]["ranks"] == 2, p.stderr[-2000:]


def test_bench_itself_has_no_cpu_path():
    """The product's bench.py on this GPU-less machine: exit 2, no line -- alone and through its own launcher."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    bench = os.path.join(ROOT, "bench.py")
    p, lines = _run(bench, "--cpu-rows", "0")
    assert p.returncode == 2 and lines == [] and "no GPU visible" in p.stderr
    p, lines = _run(bench, "--gpus", "2", "--vortices", "20000")
    assert p.returncode != 0 and lines == [] and "no GPU visible" in p.stderr


def test_ranks_do_not_outlive_a_killed_parent():
    """A harness that ends `python bench.py --gpus N` with SIGKILL must not leave ranks behind on the GPUs: the launcher the
    parent started gets SIGTERM when its parent dies (PR_SET_PDEATHSIG) and takes its ranks down."""
    import signal
    p = subprocess.Popen([sys.executable, RIG, "--gpus", "2", "--vortices", "16384", "--steps", "1", "--warmup", "0", "--deadline-s", "120"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(LUDVM_BENCH_TEST_HANG="1"))

    def descendants(pid):
        out = subprocess.run(["ps", "-e", "-o", "pid=,ppid=,args="], capture_output=True, text=True).stdout
        rows = [l.split(None, 2) for l in out.splitlines() if len(l.split(None, 2)) == 3]
        kids, frontier = [], {str(pid)}
        while frontier:
            nxt = {r[0] for r in rows if r[1] in frontier}
            kids += [r for r in rows if r[1] in frontier]
            frontier = nxt
        return kids
    try:
        t0 = time.time()
        while time.time() - t0 < 60:                     # until the two ranks exist (they hang on purpose after their region)
            ranks = [r for r in descendants(p.pid) if "bench_cpu_rig.py" in r[2] and "torch.distributed.run" not in r[2]]
            if len(ranks) >= 2:
                break
            time.sleep(0.5)
        assert len(ranks) >= 2, descendants(p.pid)
        pids = [int(r[0]) for r in descendants(p.pid)]
        p.send_signal(signal.SIGKILL)
        p.wait(10)
        t0 = time.time()
        alive = pids
        while alive and time.time() - t0 < 45:
            time.sleep(0.5)
            alive = [q for q in pids if os.path.exists(f"/proc/{q}") and "Z" not in open(f"/proc/{q}/stat").read().split(")")[1].split()[0]]
        assert not alive, alive
    finally:
        if p.poll() is None:
            p.kill()


def test_a_visible_devices_list_shorter_than_the_ranks_is_refused_before_anything_starts():
    """bench.py (the product's, nccl backend) with --gpus 4 and HIP_VISIBLE_DEVICES=0: every rank would inherit the one card.
    The parent refuses at once -- no launcher, no torch import."""
    t0 = time.time()
    p, lines = _run(os.path.join(ROOT, "bench.py"), "--gpus", "4", env=_env(HIP_VISIBLE_DEVICES="0"))
    assert p.returncode == 2 and lines == [] and "needs one GPU per rank; HIP_VISIBLE_DEVICES=0 shows 1" in p.stderr
    assert "torch.distributed.run" not in p.stderr and time.time() - t0 < 10
