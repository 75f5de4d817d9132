"""GPU tier: the bench.py output contract (one JSON line with the metric, roofline and cpu_baseline
objects), on a small workload so it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

from bench_checks import assert_other_configs, assert_self_checking_config4 as _assert_self_checking_config4, assert_sweep
from conftest import ROOT

pytestmark = pytest.mark.gpu
HANG_RIG = os.path.join(ROOT, "tests", "bench_hang_rig.py")      # bench.main() on the HIP rig + the hooks that hang on purpose


def _run(*args):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines           # exactly one line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("symmetric", [1, 0])
def test_bench_json_contract(symmetric):
    d = _run("--vortices", "40000", "--steps", "3", "--warmup", "1", "--cpu-rows", "64", "--symmetric", str(symmetric))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["kernel_variant"] == ("symmetric" if symmetric else "direct")
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and r["kernel_launches_timed"] == 3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    ex = r["executed"]       # what the ALU issued: 9 FLOP per ordered pair in the symmetric kernel, 13 in the direct one
    assert ex["flop_per_pair"] == (9 if symmetric else 13) and r["flop_per_pair"] == 13
    assert abs(ex["frac"] * 13 - r["frac"] * ex["flop_per_pair"]) < 1e-9 and ex["frac"] <= r["frac"]
    assert len(d["repeat_values"]) == 3 and all(v > 0 for v in d["repeat_values"])
    # whole-job value is consistent with the step time, and the kernel time is inside the step time
    assert abs(d["value"] - 40000.0**2 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert r["kernel_ms_avg"] <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "pairs/s" and c["value"] > 1e6
    assert c["gpu_vs_oracle_max_rel_err"] < 1e-5
    a = c["all_cores"]                      # the same sample by the C restatement on every core: reported beside, never instead
    assert "error" not in a and a["cores"] >= 1 and a["value"] > 1e6 and a["vs_numpy_max_rel_diff"] < 1e-12


def test_bench_line_carries_every_baseline_config():
    """the default N = 1 line (config 3 at N = 1e6, shortened here to 3 steps and without config 4's 15 s
    leg) also times config 5's flow field, config 2's full 50 000-step time_loop and the README case, each with its check."""
    d = _run("--steps", "3", "--warmup", "1", "--repeats", "0", "--cpu-rows", "256", "--cpu-budget", "4", "--cfg4-steps", "0")
    assert "config 3" in d["config"]["workload"] and d["config"]["n_vortices"] == 1_000_000 and "config4_one_gpu" not in d
    assert d["cpu_baseline"]["gpu_vs_oracle_max_rel_err"] < 1e-5
    assert_other_configs(d)
    # a run at another size, or with forced tuning, measures that one thing only
    d = _run("--vortices", "40000", "--steps", "2", "--warmup", "1", "--repeats", "0", "--cpu-rows", "0")
    assert not any(k in d for k in ("config5_flowfield", "config2_time_loop", "config1_readme"))


def test_bench_config4_shape_on_one_gpu():
    d = _run("--workload", "cfg4", "--vortices", "60000", "--steps", "2", "--warmup", "1", "--cpu-rows", "0")
    assert d["scaling"] == "strong" and "config 4" in d["config"]["workload"] and "cpu_baseline" not in d
    assert d["value"] > 1e11
    # one rank, no process group: no collective is issued (times 0), everything else is there
    _assert_self_checking_config4(d, 1, collectives_issued=False)
    assert d["config"]["collective_ms_per_rank"] == [0.0]


def test_bench_config4_direct_variant_reported():
    d = _run("--workload", "cfg4", "--vortices", "60000", "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--symmetric", "0")
    assert d["config"]["kernel_variant"] == "direct"
    _assert_self_checking_config4(d, 1, main="direct", collectives_issued=False)
    assert d["symmetric_variant"]["value"] > d["direct_variant"]["value"] * 0.8


def test_bench_budget_drops_the_other_variant_and_the_repeats_first():
    """--budget-s too small for anything but the reported region: the line still comes, says what was skipped."""
    d = _run("--workload", "cfg4", "--vortices", "60000", "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--budget-s", "1")
    assert d["value"] > 1e11 and d["repeat_values"] == [] and "budget" in d["direct_variant"]["skipped"]
    assert d["result_check"]["symmetric"]["ranks_agree"] is True and "direct" not in d["result_check"]


def test_bench_config4_through_the_library_communicator_on_one_rank():
    """The N > 1 bench path with the collective inside the library, on a one-rank communicator (real RCCL): the
    communicator is created, proven on a known sum, used by every step and destroyed; no fallback note."""
    d = _run("--workload", "cfg4", "--vortices", "60000", "--steps", "2", "--warmup", "1", "--cpu-rows", "0",
             "--collectives", "library")
    assert d["config"]["collective_note"] is None, d["config"]["collective_note"]
    assert "inside libludvm_hip.so" in d["config"]["collective"] and d["value"] > 1e11
    _assert_self_checking_config4(d, 1)          # ncclAllReduce AND ncclAllGather issued (one-rank identities), timed, checked
    assert_sweep(d, 1, ["library"])              # ... and the collective micro-sweep through the library's communicator
    assert "ONE rank" in d["collective_sweep_us"]["min_wake_rule"]
    ref = _run("--workload", "cfg4", "--vortices", "60000", "--steps", "2", "--warmup", "1", "--cpu-rows", "0",
               "--collectives", "torch")
    assert "torch.distributed" in ref["config"]["collective"] and ref["config"]["collective_note"] is None


def test_sharded_wake_collectives_on_the_real_rccl_backend():
    """One rank, backend "nccl" (= RCCL): the int64 all-reduce / fp32 all-gather calls of the sharded step with the
    layouts used at G > 1 (identities in a one-rank group) against the same step without collectives, bit for bit
    (tools/rccl_one_rank.py, in a child process so that the process group does not outlive the test)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_library_communicator_one_rank_through_the_c_abi():
    """ludvm_comm_* on the real RCCL with ONE rank (tools/comm_one_rank.py; LUDVM_COMM_FORCE=1 makes the one-rank group issue
    its collectives): the class-level sharded time loop with the in-library ncclAllReduce per step, the host all-gather of
    result blocks, and config 4's step with ncclAllReduce / ncclAllGather on device buffers -- all bit for bit what the
    same calls give without a communicator.  No torch.distributed in that process."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "comm_one_rank.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0 and "COMM_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def _launch(nproc, extra_env, *args):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *args]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines           # exactly one line on stdout, whatever the libraries print
    return json.loads(lines[0])


def test_bench_two_ranks_through_the_launcher():
    """The driver's launch line for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), rehearsed with two
    ranks sharing the one card over gloo (LUDVM_BENCH_BACKEND): config 4's sharded step, MAX-over-ranks timing, one
    JSON line from rank 0."""
    d = _launch(2, {"LUDVM_BENCH_BACKEND": "gloo"}, "--vortices", "120000", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "config 4" in d["config"]["workload"]
    assert d["config"]["collective_backend"] == "gloo" and d["config"]["kernel_variant"] == "symmetric"
    assert d["config"]["ranks"] == 2 and len(d["config"]["pair_kernel_ms_per_rank"]) == 2
    assert "all_reduce" in d["config"]["collective"]
    assert abs(d["value"] - 120000.0**2 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0
    assert "a launcher" in d["config"]["launched_by"]
    _assert_self_checking_config4(d, 2)
    assert_sweep(d, 2, ["torch"])


def _self_launch(nproc, extra_env, *args, timeout=600):
    """`python bench.py --gpus N ...` with NO launcher: bench.py starts its ranks itself.  A test that asks
    for a hang on purpose (LUDVM_BENCH_TEST_HANG*) runs tests/bench_hang_rig.py -- bench.main() on the same HIP rig plus the
    hooks that hang; bench.py itself reads no test variable."""
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    script = HANG_RIG if any(k.startswith("LUDVM_BENCH_TEST_HANG") for k in extra_env) else os.path.join(ROOT, "bench.py")
    p = subprocess.run([sys.executable, script, "--gpus", str(nproc), *args], capture_output=True, text=True,
                       env=env, timeout=timeout)
    return p, [l for l in p.stdout.splitlines() if l.strip()]


def test_bench_starts_its_own_ranks():
    """The driver's most natural command for N = 2: plain `python bench.py --gpus 2 ...` (two gloo ranks sharing the one
    card): one JSON line, two ranks."""
    p, lines = _self_launch(2, {"LUDVM_BENCH_BACKEND": "gloo"}, "--vortices", "120000", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and "bench.py itself" in d["config"]["launched_by"]
    assert abs(d["value"] - 120000.0**2 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    _assert_self_checking_config4(d, 2)
    assert_sweep(d, 2, ["torch"])


@pytest.mark.parametrize("n", [200000, 18000])
def test_bench_starts_four_ranks_with_padded_owner_blocks(n):
    """Four self-launched gloo ranks on the one card (a GPU box allows six processes on its card; the eight-rank rehearsal
    is tests/test_bench_cpu_rehearsal.py, on the CPU): 200 000 vortices -- the last rank's block is partly padding;
    18 000 -- blocks of 6144, the last rank owns nothing but padding."""
    p, lines = _self_launch(4, {"LUDVM_BENCH_BACKEND": "gloo"}, "--vortices", str(n), "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["ranks"] == 4 and len(d["config"]["pair_kernel_ms_per_rank"]) == 4
    _assert_self_checking_config4(d, 4, min_value=1e9)
    assert_sweep(d, 4, ["torch"])


def test_self_launched_ranks_print_the_deadline_line():
    """--deadline-s through the self-launch: the phase after the reported region never ends (on purpose); rank 0 prints the
    line as it stands, both ranks exit 0, so does the launcher, and the parent relays the one line."""
    import time
    t0 = time.time()
    p, lines = _self_launch(2, {"LUDVM_BENCH_BACKEND": "gloo", "LUDVM_BENCH_TEST_HANG": "1"}, "--vortices", "120000", "--steps", "2",
                            "--warmup", "1", "--deadline-s", "40", timeout=200)
    assert p.returncode == 0, p.stderr[-3000:]
    assert 35 < time.time() - t0 < 120
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "deadline" in d["incomplete"] and "hung on purpose" in d["incomplete"] and d["n_gpus"] == 2 and d["value"] > 1e10


def test_self_launched_ranks_survive_a_communicator_join_that_never_returns():
    """... and the join watchdog with two ranks: both time out, agree on it through torch, go on with torch.distributed's
    collectives on fresh engines."""
    p, lines = _self_launch(2, {"LUDVM_BENCH_BACKEND": "gloo", "LUDVM_BENCH_TEST_HANG_COMM": "1"}, "--vortices", "120000", "--steps", "2",
                            "--warmup", "1", "--collectives", "library", "--comm-init-timeout", "3")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "did not return within 3 s" in d["config"]["collective_note"] and "torch.distributed" in d["config"]["collective"]
    assert "incomplete" not in d
    _assert_self_checking_config4(d, 2)


def test_bench_config4_one_rank_on_the_real_rccl_backend_with_torch_collectives():
    """One rank, process group on the real "nccl" backend (LUDVM_BENCH_FORCE_DIST=1), torch.distributed's collectives
    forced: the all-reduce and the all-gather of both variants are issued on RCCL, timed and checked."""
    d = _launch(1, {"LUDVM_BENCH_FORCE_DIST": "1"}, "--workload", "cfg4", "--vortices", "60000", "--steps", "2", "--warmup", "1",
                "--collectives", "torch")
    assert "torch.distributed" in d["config"]["collective"] and d["config"]["ranks"] == 1
    _assert_self_checking_config4(d, 1)


def test_bench_stdout_is_one_line_with_the_rccl_backend_initialised():
    """RCCL prints a version banner on stdout when a communicator is created; the JSON line must still be the only
    thing there.  One rank, real "nccl" backend (LUDVM_BENCH_FORCE_DIST=1)."""
    d = _launch(1, {"LUDVM_BENCH_FORCE_DIST": "1"}, "--vortices", "40000", "--steps", "2", "--warmup", "1", "--cpu-rows", "32")
    assert d["n_gpus"] == 1 and "config 3" in d["config"]["workload"] and d["cpu_baseline"]["gpu_vs_oracle_max_rel_err"] < 1e-5


def test_engine_lifecycle_releases_device_memory():
    """300 create / run / destroy cycles (all precisions, both histories, overlapped march steps every third cycle):
    free device memory is the same after 300 cycles as after 20 (tools/soak_engine_lifecycle.py)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_engine_lifecycle.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert d["cycles"] == 300 and abs(d["leaked_MB"]) < 64


def _torchrun(nproc, script, extra_env, port):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", script)]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)


def test_class_level_sharding_two_ranks_on_one_card():
    """`LUDVM(..., distributed=True)` with two ranks sharing the one card over gloo (tools/dist_class_check.py): the
    sharded time loop (tile blocks of the symmetric roll-up + one integer all-reduce per step through ludvm_set_shard's
    hook, marched and per step, fp32 and hi+lo) and the sharded flow field equal the single-GPU run BIT FOR BIT; the
    sharded induced_velocity agrees to fp32 rounding."""
    p = _torchrun(2, "dist_class_check.py", {"LUDVM_DIST_BACKEND": "gloo"}, 29561)
    assert p.returncode == 0 and "DIST_OK gloo 2" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


def test_class_level_sharding_on_rccl_with_two_gpus():
    """The same check on the real backend, one rank per GPU over RCCL: runs wherever two or more GPUs are visible (the
    driver's 8-GPU node), skips on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL ranks cannot share a card)")
    p = _torchrun(2, "dist_class_check.py", {}, 29562)
    assert p.returncode == 0 and "DIST_OK nccl 2" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


def test_class_level_sharding_on_the_library_communicator_with_two_gpus():
    """... and with the library's own communicator instead of torch.distributed (LUDVM_DIST_COLLECTIVES=library): the
    launcher only starts the two processes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL ranks cannot share a card)")
    p = _torchrun(2, "dist_class_check.py", {"LUDVM_DIST_COLLECTIVES": "library"}, 29563)
    assert p.returncode == 0 and "DIST_OK library 2" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


def test_bench_two_ranks_on_rccl_with_two_gpus():
    """bench.py exactly as the driver launches it for N = 2, on RCCL: skips below two GPUs.  Default collectives (the
    library's own communicator) and torch.distributed's."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL ranks cannot share a card)")
    for sym in ("1", "0"):
        for coll in ("auto", "torch"):
            d = _launch(2, {}, "--vortices", "200000", "--steps", "2", "--warmup", "1", "--symmetric", sym, "--collectives", coll)
            assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "nccl" and d["config"]["ranks"] == 2
            assert len(d["config"]["pair_kernel_ms_per_rank"]) == 2 and d["value"] > 1e11
            assert ("libludvm_hip" in d["config"]["collective"]) == (coll == "auto")
            _assert_self_checking_config4(d, 2, main="symmetric" if sym == "1" else "direct")
            assert_sweep(d, 2, ["library", "torch"] if coll == "auto" else ["torch"])
    # and as the driver may well type it: no launcher
    p, lines = _self_launch(2, {}, "--vortices", "200000", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    assert json.loads(lines[0])["config"]["ranks"] == 2


def test_bench_deadline_prints_what_has_been_measured():
    """--deadline-s: a phase that never ends (here: on purpose, right after the reported region) must not take the measurement
    with it -- at the deadline rank 0 prints the line as it stands, marked `incomplete`, and the process exits 0."""
    import time
    env = dict(os.environ, LUDVM_BENCH_TEST_HANG="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, HANG_RIG, "--vortices", "40000", "--steps", "3", "--warmup", "1",
                        "--cpu-rows", "0", "--deadline-s", "25"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert 20 < time.time() - t0 < 60
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "deadline" in d["incomplete"] and "hung on purpose" in d["incomplete"]
    assert d["value"] > 1e11 and d["steps"] == 3 and d["roofline"]["kernel_launches_timed"] == 3 and d["repeat_values"] == []
    assert "deadline of 25 s reached" in p.stderr


def test_bench_survives_a_communicator_join_that_never_returns():
    """The library's communicator comes up inside ncclCommInitRank, a collective with no timeout of its own.  bench.py joins in a
    helper thread: a join that does not return within --comm-init-timeout (here: on purpose) leaves the stuck call its
    context, the run goes on with torch.distributed's collectives on a fresh engine, says so, and exits cleanly."""
    env = dict(os.environ, LUDVM_BENCH_TEST_HANG_COMM="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, HANG_RIG, "--workload", "cfg4", "--vortices", "60000", "--steps", "2",
                        "--warmup", "1", "--cpu-rows", "0", "--collectives", "library", "--comm-init-timeout", "3"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "did not return within 3 s" in d["config"]["collective_note"] and "torch.distributed" in d["config"]["collective"]
    assert "incomplete" not in d and d["value"] > 1e11
    _assert_self_checking_config4(d, 1, collectives_issued=False)


def test_bench_leaves_a_communicator_alone_that_joined_late():
    """a join that is merely SLOW returns after the timeout, when the run has moved on to
    torch.distributed's collectives on a fresh engine.  The helper thread is told (`abandoned`): it issues no all-reduce on
    the communicator nobody waits for.  Here: comm_init returns, then the hook sleeps past the timeout."""
    env = dict(os.environ, LUDVM_BENCH_TEST_HANG_COMM="late")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, HANG_RIG, "--workload", "cfg4", "--vortices", "60000", "--steps", "2",
                        "--warmup", "1", "--cpu-rows", "0", "--collectives", "library", "--comm-init-timeout", "3"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "did not return within 3 s" in d["config"]["collective_note"] and "torch.distributed" in d["config"]["collective"]
    assert "incomplete" not in d and d["value"] > 1e11
    _assert_self_checking_config4(d, 1, collectives_issued=False)


def test_bench_falls_back_when_the_first_collective_on_the_library_communicator_raises():
    """the known-sum proof runs on a side stream; if that first all-reduce RAISES (an RCCL error is what the torch
    fallback exists for) the engine must be handed back to the stream the run measures on -- try / finally -- and the run goes
    on with torch.distributed's collectives on the SAME engine (the join returned: nothing is abandoned).  The result checks of
    both variants pass."""
    env = dict(os.environ, LUDVM_BENCH_TEST_HANG_COMM="raise")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, HANG_RIG, "--workload", "cfg4", "--vortices", "60000", "--steps", "2",
                        "--warmup", "1", "--cpu-rows", "0", "--collectives", "library", "--comm-init-timeout", "30"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "raised on purpose" in d["config"]["collective_note"] and "torch.distributed" in d["config"]["collective"]
    assert "incomplete" not in d and d["value"] > 1e11
    _assert_self_checking_config4(d, 1, collectives_issued=False)


def test_the_measurement_program_reads_no_test_variable():
    """the hang hooks live in the test rigs (tests/bench_hang_hooks.py), not in bench.py."""
    assert "LUDVM_BENCH_TEST" not in open(os.path.join(ROOT, "bench.py")).read()


def test_rccl_ranks_that_would_share_a_card_are_refused_with_a_message():
    """`python bench.py --gpus 2` on a ONE-GPU box, default (nccl = RCCL) backend: two ranks cannot share a card.  Refused with
    exit 2 and a message -- by the parent when a *_VISIBLE_DEVICES list (which every rank would inherit) is shorter than N, by
    each rank before any process group is built otherwise; no hang, no RCCL error trace, no line."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    t0 = __import__("time").time()
    p, lines = _self_launch(2, {}, "--vortices", "120000", "--steps", "2", "--warmup", "1", timeout=200)
    assert p.returncode != 0 and lines == [] and "needs one GPU per rank" in p.stderr, p.stderr[-2000:]
    assert __import__("time").time() - t0 < 120
