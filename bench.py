#!/usr/bin/env python3
"""Headline benchmark: Biot-Savart pair-interactions/s of the all-pairs vortex-induction kernel.

    python bench.py                       # 1 GPU, config 3: N = 1e6 synthetic wake, all-pairs call
    python bench.py --gpus G              # config 4 on G GPUs: starts its own G ranks (a child `python -m
                                          # torch.distributed.run ...` on 127.0.0.1 and a free port) and relays their line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus G --steps K --warmup W      # the same under a launcher (RANK / WORLD_SIZE set)

A "step" is one pass of the hot path over the synthetic wake held in HBM:
  * 1 GPU  (BASELINE config 3): one induced_velocity all-pairs call, N sources x N targets;
  * G GPUs (BASELINE config 4): one self-advection step of the N = 8e6 wake, sharded over the ranks with ONE
    collective per step (ludvm_amd/sharded.py): symmetric variant (default) -- every rank evaluates its I-tile block
    of the unordered pairs, one RCCL all-reduce of the 64-bit fixed-point sums, replicated Euler update; direct
    variant (--symmetric 0) -- all N sources on the rank's own N/G targets, Euler update, one RCCL all-gather of the
    positions.  Total work per step is fixed (N^2 pairs): strong scaling.
value = ordered pair interactions (self pairs count) of all ranks / wall time, max over ranks.

WHAT THE N = 1 AND THE N > 1 LINES COMPARE.  `value` at N = 1 is config 3 (N = 1e6: the configuration the metric is quoted
on); `value` at N > 1 is config 4 (N = 8e6: the configuration that is sharded).  The same-work denominator for "8 GPUs vs 1"
is therefore NOT the N = 1 line's `value` but its `config4_one_gpu.value` (config 4's step on the one GPU); every N > 1 line
says so in `scaling_denominator`.

Order of a run: CPU baseline (rank 0, N = 1 only) -> warmup -> the timed region of exactly --steps steps (barrier +
synchronize on both sides) -> --repeats further regions of the same length, reported as `repeat_values` (box-to-box and
run-to-run spread; they also keep the GPU busy long enough for a 5-second utilisation sampler to see the run) -> at
N = 1, --cfg4-steps steps of config 4's workload (N = 8e6) on the one GPU, reported as `config4_one_gpu`, then one leg per
remaining BASELINE config, each outside `value` and each with its own check: `config5_flowfield` (4096 x 4096 grid over the
N = 1e6 wake + vorticity: ms, pairs/s, credited fraction of the fp32 peak, 256 sampled grid points against the C oracle),
`config2_time_loop` (the full 50 000-step LUDVM(dt=1e-3, tf=50): wall, roll-up pairs, pairs/s of the wall, pair-kernel seconds,
final wake, first LEV step and max |dCl| over the first 600 steps against the reference's own first 1500 steps,
tests/golden/g7_config2_first1500.npz) and `config1_readme` (the README case's wall time and `cpu_baseline_time_loop`: the same
case by the oracle's restatement of the reference's time_loop on one host core).  ~25 s on top of the rest (--cfg5/2/1 0 skip).

Config 4 (every N > 1 run) is a self-checking measurement, because nobody gets to debug it on the 8-GPU node:
  * `config.collective_ms_per_rank` beside `config.pair_kernel_ms_per_rank` (HIP events on the launch stream around the
    collective alone: transfer + wait for the slowest rank's kernel), with max / mean of both;
  * after the timed region the OTHER step variant runs too -- `symmetric_variant` (one all-reduce of the fixed-point sums,
    the default) and `direct_variant` (the north star's wording: targets in blocks + one all-gather of the positions) are
    both in the one line, each with value / ms_per_step / kernel and collective ms per rank;
  * `result_check` (outside every timed region), per variant: one more step with a power-of-two dt of 2^16, so that the
    displacement IS the velocity to 1e-7 -- (a) every rank's positions reduced to a 64-bit checksum and compared across
    ranks (`ranks_agree`), (b) 256 sampled displacements of rank 0 against the float64 C oracle on the positions before
    that step (`gpu_vs_oracle_max_rel_err`, relative to max|u|);
  * `collective_sweep_us` (after the checks, < 1 s): the collectives of the CLASS-level sharding at its threshold sizes --
    int64 all-reduce of 0.5 / 1 / 2 / 4 / 8 MB (a sharded roll-up step of 32 768 ... 524 288 vortices) and all-gather of
    8 B x {65 536, 262 144, 1e6} / G per rank, 3 warm-ups + 20 timed repetitions each on the stream the step uses, max over
    ranks -- and `min_wake_suggested`: the smallest wake whose all-reduce costs less than half of what the split saves
    there (profiles/r04_shard_break_even.txt).  The first N > 1 run thereby measures ludvm_amd/comm.py's MIN_WAKE.
  * a time budget (--budget-s, default 300 s of wall time for the whole process): the repeats and the other variant's
    step count shrink to fit (N = 2 steps take 3.5 s / 5.5 s each); what was run is stated (`steps`).
  * two safety nets, because a hang in an optional late phase must not lose the measurement: the library's communicator is
    joined in a helper thread (--comm-init-timeout: past it torch.distributed's collectives take over on a fresh engine,
    and the helper is told it has been abandoned: it issues nothing more); and past --deadline-s (540 s) rank 0 prints the
    line as it stands -- `incomplete` names the phase that did not finish -- and every rank exits.  The line is rebuilt at
    every milestone (reported region, other variant, each repeat, checks, sweep).
    Exit status: 0 with a complete line; 0 with a line marked `incomplete` (the measurement is the point: under a launcher
    one non-zero rank would make the launcher discard rank 0's line); 3 when the deadline came before anything was measured.

The collective of the N > 1 step is issued INSIDE libludvm_hip.so on its own RCCL communicator (ludvm_comm_*;
--collectives library, the default on the nccl backend; torch.distributed only ships the 128-byte identifier and does
the barrier / max-over-ranks of the contract) or by torch.distributed (--collectives torch; gloo rehearsals).

Prints ONE JSON line on rank 0's stdout (everything else that libraries print there, e.g. RCCL's version
banner, is routed to stderr).  Synthetic inputs follow SURVEY.md section 8(d):
rng = default_rng(20260101); x ~ U(-10,0), z ~ U(-2,2), Gamma ~ N(0,1)/N; v_core = 0.065.

The machine is reached through a "rig" (HipRig: the HIP engine on cuda:local_rank; there is no other in this file, and
without a GPU it exits 2).  tests/bench_cpu_rig.py injects a stand-in built on the oracle so that the control flow of the
N > 1 path -- self-launch, eight ranks, budget, checks, sweep -- is rehearsed under gloo on a machine with no GPU.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBPS = 8000.0
FLOP_PER_PAIR = 13                # algorithmic: sub sub mul fma fma rsq mul fma fma (FMA = 2, rsq = 1), SURVEY 8(d)
EXECUTED_FLOP_PER_PAIR = {"direct": 13, "symmetric": 9}   # the symmetric kernel shares dx, dz, r^2, q, rsq between (i,j), (j,i)
# HBM-side bytes per launch of the dominant kernel at config 3 (N = 1e6): FETCH_SIZE + WRITE_SIZE of separate
# rocprofv3 --pmc passes of this same command, committed under profiles/ (not collected in this run -- a PMC pass
# cannot share a run with the timing).  Keyed by the exact kernel the figure was taken from; a run whose kernel
# differs (another tile, another accumulation scheme) reports traffic = null rather than a stale constant.
# Units as the guide's HBM section prescribes: FETCH_SIZE counts 64 B per 128-B streaming request (doubled here for
# the direct kernel's coalesced reads); WRITE_SIZE is exact for the atomics (one 8-B integer per lane).
PMC_TRAFFIC_CFG3 = {
    # round 6 passes on the current code (tools/profile_batch.sh r06; tools/roofline_table.py recomputes DESIGN.md's table
    # from the same files).  2 x FETCH_SIZE 71 484 KB + WRITE_SIZE 70 313 KB (the partial slabs of the source splits)
    "ludvm::pair_f32<2,1024> direct, partial slabs": {"bytes": (2 * 71484.0 + 70312.5) * 1024,
                                                      "source": "profiles/r06_bench_cfg3_direct_pmc_{fetch,write}.csv"},
    # The quad variant (four I tiles of a workgroup share each partner tile: one fixed-point atomic per J vortex and
    # workgroup instead of one per wave) + a launch of the plain kernel for the diagonal tiles.
    # 2 x FETCH_SIZE (105 629 + 6 024) KB (gfx950 tallies the 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section)
    # + WRITE_SIZE (6.045e6 + 16 604) KB (exact for stores and 8-byte atomics: 1.024e8 + 2.8e5 64-B atomic requests,
    # memory-side, mostly Infinity-Cache resident: the accumulators are 16 MB)
    "ludvm::pair_sym_quad_f32<8> (+ pair_sym_f32<8> on the diagonal tiles), fixed-point accumulation": {
        "bytes": (2 * (105629.0 + 6024.2) + 6045030.0 + 16603.5) * 1024,
        "source": "profiles/r06_bench_cfg3_sym_pmc_{fetch,write}.csv"},
}
V_CORE = 0.065
DT = 5e-2

# What a sharded roll-up step SAVES per step (one GPU: the whole ring minus the slowest owner's tile block), microseconds,
# by wake size and number of ranks -- profiles/r04_shard_break_even.txt [MI355X] (tests/test_profiles_table.py holds this
# copy to the file).  A sharded step of n vortices all-reduces 16 n bytes.
SHARD_SAVED_US = {32768: {2: 62.3, 4: 95.3, 8: 99.3}, 65536: {2: 248.5, 4: 372.0, 8: 405.3},
                  131072: {2: 957.1, 4: 1460.1, 8: 1661.4}, 262144: {2: 3843.0, 4: 5897.2, 8: 6929.9},
                  524288: {2: 15900.7, 4: 23755.4, 8: 27881.1}}
SWEEP_GATHER_TARGETS = (65536, 262144, 1000000)      # induced_velocity targets in blocks: 8 B (u, w as fp32) per target
SWEEP_WARMUP, SWEEP_REPS = 3, 20
SCALING_DENOMINATOR = ("config4_one_gpu.value of the N = 1 line (config 4's N = 8e6 step on ONE GPU); the N = 1 line's own "
                       "`value` is config 3 (N = 1e6) and is not the same work")


def synthetic_wake(n):
    rng = np.random.default_rng(20260101)
    x = rng.uniform(-10, 0, n)
    z = rng.uniform(-2, 2, n)
    g = rng.standard_normal(n) / n
    return x.astype(np.float32), z.astype(np.float32), g.astype(np.float32)


def usable_cpus():
    """CPUs this process may run on (affinity mask / cgroup cpuset), not the machine's count."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(x, z, g, rows, budget_s):
    """The reference arithmetic as written (float64 NumPy broadcast, oracle/ludvm_oracle.py restating
    LUDVM.py:549-570) on a bounded sample: the first `rows` targets against all sources, one core.  Returns the
    record and the (u, w) it computed, which the GPU results are checked against afterwards."""
    from oracle import ludvm_oracle as O
    xs, zs, gs = x.astype(np.float64), z.astype(np.float64), g.astype(np.float64)
    per_chunk = max(1, int(2.0e7 // len(xs)))   # ~160 MB per [rows, N] float64 temporary
    done, t0 = 0, time.perf_counter()
    u = np.empty(rows)
    w = np.empty(rows)
    while done < rows and (time.perf_counter() - t0 < budget_s or done == 0):
        b = min(rows, done + per_chunk)
        u[done:b], w[done:b] = O.induced_velocity(gs, xs, zs, xs[done:b], zs[done:b], V_CORE)
        done = b
    el = time.perf_counter() - t0
    rec = {"value": done * len(xs) / el, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": f"first {done} targets x all {len(xs)} sources, float64 NumPy broadcast as the reference "
                     f"writes it (LUDVM.py:549-570), {el:.1f} s; host has {os.cpu_count()} logical CPUs"}
    # ... and what the whole host can do with the same arithmetic: the C restatement (oracle/pair_oracle.c, the same operations
    # in the same order per pair, OpenMP over targets) on the same sample, every core -- reported beside the reference's own
    # single-threaded form, never instead of it ("optionally also an all-cores figure")
    try:
        from oracle import c_oracle
        c_oracle.set_threads(usable_cpus())          # (the affinity mask, not os.cpu_count(): ADVICE r5)
        # bounded like the NumPy leg: chunks of targets until the sample is done or a quarter of --cpu-budget is spent
        # (without OpenMP, or on a few cores, the whole sample would take tens of seconds)
        uc, wc = np.empty(done), np.empty(done)
        chunk = max(1, min(done, 64 * c_oracle.threads()))
        done_c, t1 = 0, time.perf_counter()
        while done_c < done and (time.perf_counter() - t1 < 0.25 * budget_s or done_c == 0):
            b = min(done, done_c + chunk)
            uc[done_c:b], wc[done_c:b] = c_oracle.induced_velocity(gs, xs, zs, xs[done_c:b], zs[done_c:b], V_CORE)
            done_c = b
        el_c = time.perf_counter() - t1
        scale = max(np.abs(u[:done]).max(), np.abs(w[:done]).max())
        rec["all_cores"] = {"value": done_c * len(xs) / el_c, "unit": "pairs/s", "cores": c_oracle.threads(), "kind": "port",
                            "sample": f"the first {done_c} of the same {done} targets x {len(xs)} sources by the C restatement, OpenMP over "
                                      f"targets, {el_c:.2f} s",
                            "vs_numpy_max_rel_diff": float(max(np.abs(uc[:done_c] - u[:done_c]).max(),
                                                               np.abs(wc[:done_c] - w[:done_c]).max()) / scale)}
    except Exception as e:       # noqa: BLE001  (an extra; the contract's baseline is the record above)
        rec["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    return rec, u[:done], w[:done]


def multi_process_gpu_environment(env=os.environ):
    """HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller chose a value (ludvm_amd.comm.prepare_ipc_environment: where the
    default comes from and why it must precede the first HIP call).  Imported lazily: ludvm_amd/comm.py imports no GPU code."""
    from ludvm_amd.comm import prepare_ipc_environment
    return prepare_ipc_environment(env, force=True)


# ======================================================================================================================
# --gpus N without a launcher: start the ranks as a child process
# ======================================================================================================================
def launch_ranks(n, argv, deadline_s):
    """`python bench.py --gpus N` with no launcher's environment (WORLD_SIZE / RANK unset): this process -- which has not
    imported torch and never touches the GPU -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD (no exec), passes rank 0's one JSON
    line through to stdout (anything else the launcher prints there goes to stderr), lets stderr through, and returns the
    child's exit status.  The child runs in a process group of its own: a signal to this process, or a child that outlives
    the run's own deadline by two minutes, ends exactly that group."""
    import signal
    import socket
    import subprocess
    if os.environ.get("LUDVM_BENCH_BACKEND", "nccl") == "nccl":
        # every rank will inherit THIS environment: a *_VISIBLE_DEVICES list shorter than N means RCCL ranks sharing a card
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            listed = [d for d in os.environ.get(var, "").split(",") if d.strip()]
            if listed and len(listed) < n:
                print(f"bench.py: --gpus {n} on the nccl (RCCL) backend needs one GPU per rank; {var}={os.environ[var]} shows "
                      f"{len(listed)} (LUDVM_BENCH_BACKEND=gloo rehearses several ranks on one card)", file=sys.stderr)
                return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0]), *argv]
    env = dict(os.environ, LUDVM_BENCH_SELF_LAUNCHED="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):          # (the launcher sets them per rank)
        env.pop(k, None)
    multi_process_gpu_environment(env)                      # (the ranks set it again themselves: ludvm_amd/comm.py has the why)
    print(f"bench.py: --gpus {n} without a launcher: starting {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)

    def die_with_parent():
        # a harness that ends THIS process with SIGKILL (no handler runs) must not leave eight ranks on the GPUs: the launcher
        # gets SIGTERM when its parent dies and takes its ranks down (Linux prctl PR_SET_PDEATHSIG = 1)
        try:
            import ctypes
            ctypes.CDLL(None).prctl(1, int(signal.SIGTERM), 0, 0, 0)
        except Exception:       # noqa: BLE001
            pass
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, preexec_fn=die_with_parent)

    def end_group(sig=signal.SIGTERM):
        try:
            os.killpg(child.pid, sig)            # the group we created (start_new_session): the launcher and its ranks
        except ProcessLookupError:
            pass

    def on_signal(signum, _frame):
        end_group(signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    limit = (deadline_s + 120.0) if deadline_s > 0 else None
    if limit is not None:
        def overrun():
            print(f"bench.py: the ranks are still running {limit:.0f} s after the start; ending them", file=sys.stderr, flush=True)
            end_group()
            time.sleep(10.0)
            end_group(signal.SIGKILL)
        timer = threading.Timer(limit, overrun)
        timer.daemon = True
        timer.start()
    lines = 0
    for raw in child.stdout:
        text = raw.decode(errors="replace")
        is_line = False
        if lines == 0 and text.lstrip().startswith("{"):
            try:
                is_line = "metric" in json.loads(text)
            except ValueError:
                is_line = False
        if is_line:
            lines += 1
            sys.stdout.write(text if text.endswith("\n") else text + "\n")
            sys.stdout.flush()
        else:
            sys.stderr.write(text)
    rc = child.wait()
    if limit is not None:
        timer.cancel()
    if rc == 0 and lines == 0:
        print("bench.py: the ranks exited 0 without a line", file=sys.stderr)
        return 3
    return rc if rc >= 0 else 128 - rc


# ======================================================================================================================
# the machine
# ======================================================================================================================
class HipRig:
    """What the benchmark needs of the machine: a device, the engine, the pair arithmetic of a shard step, a fence and a
    stopwatch on the launch stream.  The HIP engine on cuda:<local_rank>; no GPU -> exit 2 (there is no CPU path)."""
    name = "hip"
    default_backend = "nccl"

    def __init__(self, local_rank):
        import torch
        if not torch.cuda.is_available():
            print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
            sys.exit(2)
        self.torch = torch
        # modulo the visible devices: also right when a launcher shows each rank only its own card
        self.dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.dev_index)
        self.device = torch.device("cuda", self.dev_index)

    def new_engine(self):
        from ludvm_amd import Engine
        eng = Engine(self.dev_index)
        eng.set_stream(self.torch.cuda.current_stream().cuda_stream)
        return eng

    def shard_kernel(self, eng):
        from ludvm_amd.sharded import HipShardKernel
        return HipShardKernel(eng)

    def bind_thread(self):
        self.torch.cuda.set_device(self.dev_index)            # (the current device is per thread)

    # Two places where a TEST rig can make the run misbehave on purpose (tests/bench_hang_rig.py, tests/bench_cpu_rig.py: a
    # phase that never ends, a communicator join that never returns or returns late).  The machine itself does nothing here.
    def at_milestone(self, reporter, phase):
        pass

    def at_comm_join(self, stage, timeout_s):
        pass

    def sync(self):
        self.torch.cuda.synchronize()

    def stopwatch(self):
        """-> (mark, elapsed_ms): mark() records an event on the launch stream (torch's current stream, which the engine
        launches on too); elapsed_ms(a, b) waits for b."""
        torch = self.torch

        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e

        def elapsed_ms(a, b):
            b.synchronize()
            return a.elapsed_time(b)
        return mark, elapsed_ms


# ======================================================================================================================
# the line, the milestones and the deadline
# ======================================================================================================================
class Reporter:
    """stdout carries exactly ONE line, the JSON.  Libraries write there too (RCCL prints a version banner when a
    communicator is created, gloo likewise), so file descriptor 1 is pointed at stderr for the whole run and the JSON line
    goes to the saved descriptor.  `line` is what has been measured so far as a finished line: replaced (never edited in
    place) at every milestone; `emit` writes one line once.  A daemon thread watches the wall clock until `finish()`:
    past --deadline-s the run is cut short with whatever line exists."""

    def __init__(self, rank, deadline_s, t_process):
        self.rank, self.deadline_s, self.t_process = rank, deadline_s, t_process
        sys.stdout.flush()
        self.json_fd = os.dup(1)
        os.dup2(2, 1)
        self.line, self.phase = None, "start-up"
        self.published = False          # a milestone has been reached (on every rank alike: they pass them in lockstep)
        self.emitted = False
        self.finished = False
        self._lock = threading.Lock()
        if deadline_s > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def emit(self, line):
        with self._lock:
            if self.emitted:
                return False
            self.emitted = True
            if self.rank == 0:
                os.write(self.json_fd, (json.dumps(line) + "\n").encode())
            return True

    def _watch(self):
        while not self.finished:
            if time.perf_counter() - self.t_process > self.deadline_s:
                if not self.emitted and self.rank == 0 and self.line is not None:
                    self.emit(dict(self.line, incomplete=f"--deadline-s {self.deadline_s:.0f} reached during: {self.phase}",
                                   wall_s=time.perf_counter() - self.t_process))
                print(f"bench.py: deadline of {self.deadline_s:.0f} s reached during: {self.phase}", file=sys.stderr, flush=True)
                # 0: a line exists (complete, or marked incomplete) -- on every rank, or a launcher would throw rank 0's line
                # away with the job; 3: nothing had been measured yet
                os._exit(0 if (self.emitted or self.published) else 3)
            time.sleep(0.5)

    def finish(self):
        self.finished = True


def check_wake(wake, g, rank, rig):
    """Result check of a config-4 wake, outside every timed region (the oracle as the CHECKER, like the cpu_baseline leg's
    gpu_vs_oracle_max_rel_err): one more step with dt = 2^16 -- u ~ 2e-5 at N = 8e6, so with the run's dt = 0.05 a
    displacement is one ulp of x and says nothing; with 2^16 it is the velocity to 1e-7 of max|u| -- then
      ranks_agree: every rank's positions as 64-bit checksums, before and after that step, equal on all ranks;
      gpu_vs_oracle_max_rel_err (rank 0): 256 sampled displacements / dt against the float64 C oracle
      (oracle/pair_oracle.c restating LUDVM.py:549-570) on the positions before the step, relative to max|u|."""
    import torch
    rec = {"check_dt": CHECK_DT, "samples": CHECK_SAMPLES}
    try:
        n = wake.n
        before = wake.gather_checksums()
        x0, z0 = wake.xs[:n].clone(), wake.zs[:n].clone()
        dt_run, wake.dt = wake.dt, CHECK_DT
        try:
            wake.step()
        finally:
            wake.dt = dt_run
        rig.sync()
        after = wake.gather_checksums()
        rec["ranks_agree"] = all(c == before[0] for c in before) and all(c == after[0] for c in after)
        rec["checksum"] = [f"{v & 0xFFFFFFFFFFFFFFFF:016x}" for v in after[0]]
        if rank == 0:
            from oracle import c_oracle          # the checker; never the thing measured
            idx = np.sort(np.random.default_rng(4).choice(n, size=min(CHECK_SAMPLES, n), replace=False))
            ti = torch.from_numpy(idx).to(wake.xs.device)
            x0h, z0h = x0.cpu().numpy().astype(np.float64), z0.cpu().numpy().astype(np.float64)
            ug = (wake.xs[:n][ti].cpu().numpy().astype(np.float64) - x0h[idx]) / CHECK_DT
            wg = (wake.zs[:n][ti].cpu().numpy().astype(np.float64) - z0h[idx]) / CHECK_DT
            t0 = time.perf_counter()
            if c_oracle.threads() < 16:          # (launchers pin OMP_NUM_THREADS to 1 per rank; the other ranks are waiting)
                c_oracle.set_threads(min(16, usable_cpus()))
            uo, wo = c_oracle.induced_velocity(np.asarray(g, dtype=np.float64), x0h, z0h, x0h[idx], z0h[idx], wake.v_core)
            scale = max(np.abs(uo).max(), np.abs(wo).max())
            err = max(np.abs(ug - uo).max(), np.abs(wg - wo).max()) / scale
            rec.update(gpu_vs_oracle_max_rel_err=float(err) if np.isfinite(err) else None, max_abs_u=float(scale),
                       finite=bool(np.isfinite(ug).all() and np.isfinite(wg).all()),
                       oracle_s=time.perf_counter() - t0, oracle_threads=c_oracle.threads())
    except Exception as e:       # noqa: BLE001  (a failed check is reported, it does not take the measurement with it)
        rec["error"] = f"{type(e).__name__}: {e}"
    return rec


CHECK_DT = 65536.0     # 2^16: dt * u is exact, and |dt u| ~ 1 dwarfs the rounding of x + dt u (half an ulp of x ~ 5e-7)
CHECK_SAMPLES = 256


def join_library_communicator(eng, rank, world, rig, backend, dist, timeout_s=90.0):
    """The engine's own communicator: rank 0's identifier goes round through torch (any channel would do); its sharding of
    resident-wake roll-ups is switched off (min_vortices), ShardedWake hands out the tile blocks itself.  If librccl cannot
    be opened (every rank fails alike, before any collective), torch's collectives take over.
    The join (ncclCommInitRank: a collective with no timeout of its own) and the known-sum proof run in a helper thread:
    a rank whose join does not return within `timeout_s` reports failure, the ranks agree on the outcome through torch, and
    the run goes on with torch.distributed's collectives on a FRESH engine (the stuck call keeps the old context).  The
    helper is told (`abandoned`): a join that was merely slow and returns after the timeout issues NO collective on the
    communicator nobody waits for -- two communicators driven from two threads beside the timed collectives is how RCCL
    deadlocks.  -> (collectives, note, engine_abandoned)"""
    torch, device = rig.torch, rig.device
    note = None
    try:
        uid = [eng.comm_unique_id() if rank == 0 else None]
    except Exception as e:       # noqa: BLE001
        uid = [None]
        note = f"library communicator unavailable on rank 0 ({e}); torch.distributed collectives used"
    if world > 1:
        dist.broadcast_object_list(uid, src=0, device=device if backend == "nccl" else None)
    if uid[0] is None:
        return "torch", (note or "library communicator unavailable on rank 0; torch.distributed collectives used"), False
    # join, then prove the communicator on a known sum before the steps depend on it; the ranks agree on the outcome through
    # torch (one rank falling back alone would leave the others inside a collective)
    res = {"ok": 0, "why": f"ludvm_comm_init did not return within {timeout_s:.0f} s"}
    abandoned = threading.Event()

    def join():
        try:
            rig.bind_thread()
            rig.at_comm_join("before", timeout_s)
            eng.comm_init(rank, world, uid[0], min_vortices=1 << 62)
            rig.at_comm_join("after", timeout_s)
            if abandoned.is_set():                      # the run has moved on without this communicator: touch nothing
                res.update(ok=0, why="joined after the timeout; left alone", late=True)
                return
            chk = torch.arange(1, 9, dtype=torch.int64, device=device) * (rank + 1)
            rig.sync()
            # the proof runs on a stream of its own: should the ranks ever disagree at the edge of the timeout (some abandon the
            # communicator, some issue this all-reduce), the kernel that waits for its peers must not sit on the stream the
            # measurement runs on
            side = torch.cuda.Stream(device=device)
            eng.set_stream(side.cuda_stream)
            try:
                rig.at_comm_join("proof", timeout_s)
                eng.comm_allreduce_i64_dev(chk.data_ptr(), chk.numel())
                side.synchronize()
            finally:
                # whatever the proof did -- an RCCL error on this first collective is what the fallback below exists for -- the
                # engine goes back to the stream the run measures on (unless the run has already moved on to a fresh engine)
                if not abandoned.is_set():
                    eng.set_stream(torch.cuda.current_stream().cuda_stream)
            want = torch.arange(1, 9, dtype=torch.int64) * (world * (world + 1) // 2)
            if torch.equal(chk.cpu(), want):
                res.update(ok=1, why="")
            else:
                res.update(ok=0, why="wrong sum from the library's all-reduce")
        except Exception as e:       # noqa: BLE001
            res.update(ok=0, why=str(e))
    th = threading.Thread(target=join, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        abandoned.set()
        th.join(1.0)                                    # (one that was just finishing)
    hung = th.is_alive() or bool(res.get("late"))       # either way the communicator is left to itself from here on
    ok, why = (0, f"ludvm_comm_init did not return within {timeout_s:.0f} s") if hung else (res["ok"], res["why"])
    flag = torch.tensor([ok, 1 if hung else 0], dtype=torch.int64, device=device)
    if world > 1:
        both = torch.stack([flag[0], -flag[1]])          # MIN of ok, MAX of hung
        dist.all_reduce(both, op=dist.ReduceOp.MIN)
        flag = torch.stack([both[0], -both[1]])
    all_ok, any_hung = int(flag[0].item()), int(flag[1].item())
    if all_ok == 0:
        if not any_hung:
            try:
                eng.comm_destroy()
            except Exception:       # noqa: BLE001
                pass
        return ("torch", f"library communicator not usable ({why or 'another rank failed'}); torch.distributed collectives used",
                bool(any_hung))
    return "library", None, False


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["auto", "cfg3", "cfg4"], default="auto")
    ap.add_argument("--vortices", type=int, default=0, help="override the number of vortices")
    ap.add_argument("--tpl", type=int, default=0, help="targets per lane (0 = engine heuristic)")
    ap.add_argument("--splits", type=int, default=0, help="source splits (0 = engine heuristic)")
    ap.add_argument("--symmetric", type=int, choices=[0, 1], default=1,
                    help="1: self-interaction launches use the symmetric kernel (each unordered pair once); 0: direct")
    ap.add_argument("--cpu-rows", type=int, default=2048, help="targets in the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work at most")
    ap.add_argument("--repeats", type=int, default=3, help="further timed regions of --steps steps after the reported one")
    ap.add_argument("--collectives", choices=["auto", "torch", "library"], default="auto",
                    help="N > 1: who issues the step's collective -- the library's own RCCL communicator (default on nccl) or torch.distributed")
    ap.add_argument("--budget-s", type=float, default=300.0,
                    help="config 4: wall-time budget of the whole process; repeats and the other variant's steps shrink to fit")
    ap.add_argument("--deadline-s", type=float, default=540.0,
                    help="hard limit on the process's wall time: past it rank 0 prints the line with what has been measured so far "
                         "(\"incomplete\" names the phase that did not finish) and every rank exits -- a hang in a late, optional "
                         "phase must not take the measurement with it (0 = no limit)")
    ap.add_argument("--comm-init-timeout", type=float, default=90.0,
                    help="config 4: seconds the library's communicator may take to come up before torch.distributed takes over")
    ap.add_argument("--other-variant", type=int, choices=[0, 1], default=1,
                    help="config 4: also time the other step variant (direct <-> symmetric) after the reported one")
    ap.add_argument("--check", type=int, choices=[0, 1], default=1,
                    help="config 4: result check after the timed regions (cross-rank checksum + sampled oracle check)")
    ap.add_argument("--sweep", type=int, choices=[0, 1], default=1,
                    help="config 4: collective micro-sweep at the class-level sharding's threshold sizes after the checks (< 1 s)")
    ap.add_argument("--cfg5", type=int, choices=[0, 1], default=1, help="N = 1: config 5's flow field (4096^2 x 1e6, ~2 s per call) after config 4's leg")
    ap.add_argument("--cfg2", type=int, choices=[0, 1], default=1, help="N = 1: config 2's full 50 000-step time_loop (~11 s)")
    ap.add_argument("--cfg1", type=int, choices=[0, 1], default=1, help="N = 1: config 1 (README case) + its CPU baseline by the oracle (~5 s of CPU)")
    ap.add_argument("--cfg4-steps", type=int, default=2,
                    help="N = 1: steps of config 4's workload (N = 8e6, ~7 s each) timed on the one GPU after the config-3 run (0 = skip)")
    return ap.parse_args(argv)


class Run:
    """One rank's run: arguments, the rig, the process group, the engine and the helpers every phase uses."""

    def __init__(self, args, rig_factory, t_process):
        self.args, self.t_process = args, t_process
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.rep = Reporter(self.rank, args.deadline_s, t_process)
        if self.world != args.gpus:
            if self.rank == 0:
                print(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {self.world}", file=sys.stderr)
            sys.exit(2)
        if self.world > 1:
            multi_process_gpu_environment()       # before torch is imported and before the first HIP call of this process
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rig = rig = rig_factory(self.local_rank)
        self.device = rig.device
        # Rehearsal on a one-GPU box (not a measurement): LUDVM_BENCH_BACKEND=gloo lets several ranks share
        # a card so the multi-rank control flow can be exercised; the driver's runs use RCCL, one GPU each.
        self.backend = os.environ.get("LUDVM_BENCH_BACKEND", rig.default_backend)
        # LUDVM_BENCH_FORCE_DIST=1: create the process group even for one rank (tests: the real RCCL backend on a
        # one-GPU box); the single-rank workload and its code path are unchanged
        self.force_dist = self.world == 1 and os.environ.get("LUDVM_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
        if self.world > 1 or self.force_dist:
            if self.backend == "nccl" and rig.name == "hip":
                # RCCL ranks cannot share a card: say so before the process group is built (LUDVM_BENCH_BACKEND=gloo rehearses
                # several ranks on one card).  One visible device per rank is fine: the launcher isolated the cards.
                ndev, local_world = torch.cuda.device_count(), int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
                if 1 < ndev < local_world or (ndev == 1 and local_world > 1 and "ROCR_VISIBLE_DEVICES" not in os.environ
                                              and "HIP_VISIBLE_DEVICES" not in os.environ and "CUDA_VISIBLE_DEVICES" not in os.environ):
                    if self.rank == 0:
                        print(f"bench.py: --gpus {self.world} on the nccl (RCCL) backend needs one GPU per rank; {ndev} visible "
                              f"for {local_world} local ranks", file=sys.stderr)
                    sys.exit(2)
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.device)
            else:
                dist.init_process_group(backend=self.backend)
        if self.force_dist:
            t1 = torch.ones(4, dtype=torch.float32, device=self.device)
            dist.all_reduce(t1)                       # builds the communicator (and prints whatever RCCL prints)
            rig.sync()
        self.workload = args.workload if args.workload != "auto" else ("cfg3" if self.world == 1 else "cfg4")
        self.n = args.vortices or (1_000_000 if self.workload == "cfg3" else 8_000_000)
        self.x, self.z, self.g = synthetic_wake(self.n)
        # the symmetric kernel serves self-interaction launches (configs 3 and 4)
        self.symmetric = bool(args.symmetric) and self.n >= 16384
        self.variant = "symmetric" if self.symmetric else "direct"
        self.M = {}          # the reported region: elapsed, kernel_ms, launches, pairs_per_step, pairs_per_launch, desc, ...
        self.cpu_rec = self.cpu_u = self.cpu_w = None
        self.eng = None
        self.coll = None

    def start_engine(self):
        eng = self.rig.new_engine()
        eng.set_tuning(self.args.tpl, self.args.splits)
        return eng

    # ---- agreement between the ranks ----------------------------------------------------------------------------------------
    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.rig.sync()

    def over_ranks(self, v, op="max"):
        """One number agreed by all ranks (decisions about step counts must be the same everywhere)."""
        if self.world == 1:
            return float(v)
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.MIN)
        return float(t.item())

    def per_rank(self, v):
        if self.world == 1:
            return [float(v)]
        torch = self.torch
        allv = torch.zeros([self.world], dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(allv, torch.tensor([float(v)], dtype=torch.float64, device=self.device))
        return [float(q) for q in allv.cpu()]

    def left(self):
        """Seconds of the budget still unspent (the slowest rank's clock: every rank must decide alike)."""
        return self.args.budget_s - self.over_ranks(time.perf_counter() - self.t_process)

    def timed_region(self, step_fn, steps, wake=None):
        eng = self.eng
        eng.kernel_timing(True)
        eng.kernel_time_ms(reset=True)
        if wake is not None:
            wake.collective_timing(True)
            wake.collective_time_ms(reset=True)
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        self.fence()
        el = time.perf_counter() - t0
        kms, nl = eng.kernel_time_ms(reset=True)
        eng.kernel_timing(False)
        cms, nc = (0.0, 0)
        if wake is not None:
            cms, nc = wake.collective_time_ms(reset=True)
            wake.collective_timing(False)
        return self.over_ranks(el), kms, nl, cms, nc

    # ---- the JSON line, from what is known once the reported timed region is over (later phases add to it) ----------------
    def make_out(self, repeats, extra, cfg4_rec):
        from ludvm_amd.comm import MIN_PAIRS, MIN_TARGETS, MIN_WAKE
        args, M, n, world = self.args, self.M, self.n, self.world
        elapsed, kernel_ms, launches = M["elapsed"], M["kernel_ms"], M["launches"]
        pairs_per_step, pairs_per_launch = M["pairs_per_step"], M["pairs_per_launch"]
        per_rank_ms, per_rank_coll_ms = M["per_rank_ms"], M["per_rank_coll_ms"]
        ns_l, nt_l = M["ns_l"], M["nt_l"]
        symmetric, variant = self.symmetric, self.variant
        value = pairs_per_step * args.steps / elapsed
        kern_s = kernel_ms * 1e-3
        exe = EXECUTED_FLOP_PER_PAIR[variant]
        alg_tflops = FLOP_PER_PAIR * pairs_per_launch / kern_s / 1e12 if kern_s > 0 else 0.0
        exe_tflops = exe * pairs_per_launch / kern_s / 1e12 if kern_s > 0 else 0.0
        # algorithmic HBM bytes per launch: 12 B per source read, 8 B per target read, 8 B written
        alg_bytes = 12.0 * ns_l + 16.0 * nt_l
        quad = symmetric and ns_l > 639 * 512     # (the library's rule)
        kernel_name = (("ludvm::pair_sym_quad_f32<8> (+ pair_sym_f32<8> on the diagonal tiles), fixed-point accumulation" if quad
                        else "ludvm::pair_sym_f32<8> fixed-point accumulation") if symmetric
                       else "ludvm::pair_f32<2,1024> direct, partial slabs")
        traffic = PMC_TRAFFIC_CFG3.get(kernel_name) if (self.workload == "cfg3" and n == 1_000_000 and not args.tpl and not args.splits) else None
        mean = lambda v: sum(v) / len(v)      # noqa: E731
        dist_on = world > 1 or self.force_dist
        out = {
            "metric": "biot_savart_pair_interactions_per_s", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeat_values": [pairs_per_step * args.steps / r for r in repeats],
            "config": {"workload": M["desc"], "collective_backend": self.backend if world > 1 else None,
                       "ranks": self.dist.get_world_size() if dist_on else 1, "collective": M["collective"],
                       "collective_note": M["coll_note"],
                       # the IPC mode the ranks' HSA runtimes were started with (N > 1: ludvm_amd/comm.py::prepare_ipc_environment)
                       "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                       "launched_by": ("bench.py itself (python bench.py --gpus N: a child torch.distributed.run)"
                                       if os.environ.get("LUDVM_BENCH_SELF_LAUNCHED") == "1" else
                                       ("a launcher (RANK / WORLD_SIZE in the environment)" if "WORLD_SIZE" in os.environ else "python")),
                       "n_vortices": n, "v_core": V_CORE, "device": self.info["name"],
                       "cu_count": self.info["cu_count"], "kernel_variant": variant, "targets_per_lane": args.tpl or "auto",
                       "source_splits": args.splits or "auto", "pair_kernel_ms_per_rank": per_rank_ms,
                       "pair_kernel_ms_max_over_mean": max(per_rank_ms) / mean(per_rank_ms) if mean(per_rank_ms) > 0 else None,
                       "collective_ms_per_rank": per_rank_coll_ms,
                       "collective_ms_max_over_mean": (max(per_rank_coll_ms) / mean(per_rank_coll_ms)
                                                       if per_rank_coll_ms and mean(per_rank_coll_ms) > 0 else None),
                       # thresholds of the CLASS-level sharding (LUDVM(distributed=...): time_loop / induced_velocity), not
                       # used by this workload; what the collectives cost at those sizes on THIS machine is
                       # `collective_sweep_us` (N > 1 lines), and `min_wake_suggested` what follows for min_wake
                       "class_sharding_thresholds": {"min_wake": MIN_WAKE, "min_targets": MIN_TARGETS, "min_pairs": MIN_PAIRS}},
            "roofline": {
                # per the metric's definition: algorithmic FLOPs -- 13 per ordered pair -- over the
                # dominant kernel's own time.  The symmetric kernel EXECUTES 9 per ordered pair (it shares dx, dz, r^2,
                # q and the rsqrt between (i,j) and (j,i)): `executed` is the share of the vector ALU's peak that was
                # actually issued, and the one to read as hardware utilisation
                "bound": "valu",
                "kernel": kernel_name + (" (each unordered pair once; packed fp32 vector ALU; no MFMA, not HBM-bound)"
                                         if symmetric else " (packed fp32 vector ALU; no MFMA, not HBM-bound)"),
                "achieved": alg_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": alg_tflops / FP32_VECTOR_PEAK_TFLOPS, "flop_per_pair": FLOP_PER_PAIR,
                "executed": {"flop_per_pair": exe, "achieved": exe_tflops, "frac": exe_tflops / FP32_VECTOR_PEAK_TFLOPS},
                "pairs_per_launch": pairs_per_launch,
                "kernel_ms_avg": kernel_ms, "kernel_launches_timed": launches,
                "hbm_algorithmic_bytes_per_launch": alg_bytes,
                "hbm_achieved_gbps": alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0, "hbm_peak_gbps": HBM_PEAK_GBPS,
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": (traffic["source"] + " (separate rocprofv3 --pmc passes of this command; not collected "
                                   "in this run)") if traffic else "no PMC pass on record for this kernel / size",
            },
        }
        if self.workload == "cfg4":
            out["scaling_denominator"] = SCALING_DENOMINATOR
        out.update(extra)
        if cfg4_rec is not None:
            out["config4_one_gpu"] = cfg4_rec
        if self.cpu_rec is not None:
            rec = dict(self.cpu_rec)
            u_first, w_first = M.get("u_first"), M.get("w_first")
            if u_first is not None:
                scale = max(np.abs(self.cpu_u).max(), np.abs(self.cpu_w).max())
                rec["gpu_vs_oracle_max_rel_err"] = float(max(np.abs(u_first - self.cpu_u).max(),
                                                             np.abs(w_first - self.cpu_w).max()) / scale)
            else:
                rec["gpu_vs_oracle_max_rel_err"] = None
            out["cpu_baseline"] = rec
        out["wall_s"] = time.perf_counter() - self.t_process
        return out

    def publish(self, phase, repeats=(), extra=None, cfg4_rec=None):
        """A milestone: from here on the deadline prints at least this much; `phase` is what runs next."""
        if self.rank == 0:
            self.rep.line = self.make_out(list(repeats), extra or {}, cfg4_rec)
        self.rep.published = True
        self.rep.phase = phase
        self.rig.at_milestone(self.rep, phase)


# ======================================================================================================================
# config 3: one all-pairs call per step on one GPU
# ======================================================================================================================
def run_config3(R):
    args, rig, torch, eng, n = R.args, R.rig, R.torch, R.eng, R.n
    from ludvm_amd.sharded import ShardedWake
    dx, dz, dg = (torch.from_numpy(a).to(R.device) for a in (R.x, R.z, R.g))
    du, dw = torch.empty_like(dx), torch.empty_like(dx)

    def step():
        eng.induce_dev(dx.data_ptr(), dz.data_ptr(), dg.data_ptr(), n, dx.data_ptr(), dz.data_ptr(), n, V_CORE,
                       du.data_ptr(), dw.data_ptr())
    pairs_per_step = float(n) * float(n)
    desc = f"config 3: synthetic wake N={n}, one induced_velocity all-pairs call per step (targets = sources)"
    R.rep.phase = "warm-up and the reported timed region (config 3)"
    for _ in range(args.warmup):
        step()
    R.fence()
    elapsed, kernel_ms, launches, _, _ = R.timed_region(step, args.steps)
    R.M.update(elapsed=elapsed, kernel_ms=kernel_ms, launches=launches, pairs_per_step=pairs_per_step,
               pairs_per_launch=pairs_per_step, desc=desc, collective=None, coll_note=None, per_rank_ms=[kernel_ms],
               per_rank_coll_ms=None, ns_l=n, nt_l=n)
    if R.cpu_rec is not None:
        R.M["u_first"] = du[: len(R.cpu_u)].cpu().numpy().astype(np.float64)
        R.M["w_first"] = dw[: len(R.cpu_u)].cpu().numpy().astype(np.float64)
    R.publish("the repeat regions (config 3)")
    repeats = []
    for _ in range(max(0, args.repeats)):
        repeats.append(R.timed_region(step, args.steps)[0])
        R.publish("the repeat regions (config 3)", repeats)

    # N = 1: config 4's workload (N = 8e6, what --gpus 2, 4, 8 run sharded) on this one GPU, so that "8 vs 1" compares the
    # same work; not part of `value`
    cfg4_rec = None
    if R.world == 1 and args.cfg4_steps > 0 and not args.vortices:
        R.publish("config 4's workload on the one GPU (config4_one_gpu)", repeats)
        n4 = 8_000_000
        x4, z4, g4 = synthetic_wake(n4)
        wake4 = ShardedWake(x4, z4, g4, V_CORE, DT, rig.shard_kernel(eng), R.device, symmetric=R.symmetric)
        rig.sync()
        el4, k4, _, _, _ = R.timed_region(wake4.step, args.cfg4_steps)
        cfg4_rec = {"workload": f"config 4 on ONE GPU: synthetic wake N={n4}, one self-advection step per step ({R.variant} kernel, "
                                "no collective)", "value": wake4.pairs_per_step * args.cfg4_steps / el4, "unit": "pairs/s",
                    "steps": args.cfg4_steps, "warmup": 0, "ms_per_step": el4 / args.cfg4_steps * 1e3, "pair_kernel_ms": k4,
                    "role": "the same-work denominator of the N = 2, 4, 8 lines (their `scaling_denominator`)"}
        if args.check:
            R.publish("the result check of config4_one_gpu", repeats, None, dict(cfg4_rec))
            cfg4_rec["result_check"] = check_wake(wake4, g4, R.rank, rig)
        del wake4
    # ... and the other BASELINE configs, each with its own check: 5 (flow field), 2 (full time_loop), 1 (README)
    extra = {}
    if R.world == 1 and not args.vortices and R.workload == "cfg3" and not args.tpl and not args.splits:
        extra = other_configs(R, dx, dz, dg, repeats, cfg4_rec)
    return repeats, extra, cfg4_rec



# ======================================================================================================================
# N = 1: a driver-timed figure for the other BASELINE configs (5, 2, 1) -- outside `value`, after config 4's leg
# ======================================================================================================================
README_CASE = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")   # config 1


def leg_config5(R, dx, dz, dg):
    """BASELINE config 5: flowfield 4096 x 4096 grid over the N = 1e6 synthetic wake (LUDVM.flowfield, LUDVM.py:1186-1298:
    grid targets + vorticity stencil, on the device), one untimed call, one timed; 256 sampled grid points against the float64
    C oracle (the checker), bound 1e-5 of max|u| (the test's: tests/test_gpu_kernel.py::test_full_size_config5_flowfield)."""
    torch, eng, rig, n = R.torch, R.eng, R.rig, R.n
    nx = nz = 4096
    xmin, zmin, dr = -8.0, -4.0, 8.0 / nx
    du = torch.empty(nx * nz, dtype=torch.float32, device=R.device)
    dw, dome = torch.empty_like(du), torch.empty_like(du)

    def run():
        eng.flowfield_dev(xmin, zmin, dr, nx, nz, dx.data_ptr(), dz.data_ptr(), dg.data_ptr(), n, V_CORE, du.data_ptr(), dw.data_ptr())
        eng.vorticity_dev(du.data_ptr(), dw.data_ptr(), nx, nz, dr, dome.data_ptr())
    run()
    el, kms, nl, _, _ = R.timed_region(run, 1)
    pairs = float(nx) * nz * n
    rec = {"workload": f"config 5: flowfield {nx} x {nz} grid over the N={n} synthetic wake + vorticity stencil, fp32 (4 x 4 patch kernel)",
           "ms": el * 1e3, "value": pairs / el, "unit": "pairs/s", "steps": 1, "warmup": 1, "pair_kernel_ms": kms,
           "pair_kernel_launches": nl,
           "credited_frac": FLOP_PER_PAIR * pairs / (kms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS if kms > 0 else None,
           "omega_finite": bool(torch.isfinite(dome).all().item())}
    from oracle import c_oracle          # the checker; never the thing measured
    sel = np.sort(np.random.default_rng(5).choice(nx * nz, CHECK_SAMPLES, replace=False))
    xt, zt = xmin + (sel // nz) * dr, zmin + (sel % nz) * dr
    if c_oracle.threads() < 16:
        c_oracle.set_threads(min(16, usable_cpus()))
    uo, wo = c_oracle.induced_velocity(R.g.astype(np.float64), R.x.astype(np.float64), R.z.astype(np.float64), xt, zt, V_CORE)
    st = torch.from_numpy(sel).to(R.device)
    ug, wg = du[st].cpu().numpy().astype(np.float64), dw[st].cpu().numpy().astype(np.float64)
    scale = max(np.abs(uo).max(), np.abs(wo).max())
    rec.update(samples=CHECK_SAMPLES, gpu_vs_oracle_max_rel_err=float(max(np.abs(ug - uo).max(), np.abs(wg - wo).max()) / scale),
               bound=1e-5)
    return rec


def leg_config2(R):
    """BASELINE config 2: NACA0012 sinusoidal pitch, dt = 1e-3, t in [0, 50]: the full 50 000-step LUDVM.time_loop
    (LUDVM.py:597-1171) in fp32 on a fresh engine, device-resident march, sparse history.  Checked against the REFERENCE's own
    run of the first 1500 steps (tests/golden/g7_config2_first1500.npz, generated by importing the reference): first LEV at
    step 1335, max |dCl| over the first 600 steps <= 1e-5 (the fp32 tier of tests/test_gpu_wake.py; chaos beyond ~1000)."""
    from ludvm_amd import LUDVM
    eng = R.rig.new_engine()
    eng.kernel_timing(True)
    eng.kernel_time_ms(reset=True)
    R.fence()
    t0 = time.perf_counter()
    sim = LUDVM(**dict(README_CASE, dt=1e-3, tf=50), verbose=False, engine=eng, precision="f32", history="sparse", snapshot_steps=[],
                run=False)
    t_setup = time.perf_counter() - t0
    sim.time_loop()
    sim.compute_coefficients()
    R.fence()
    wall = time.perf_counter() - t0
    kms, nl = eng.kernel_time_ms(reset=True)
    eng.kernel_timing(False)
    shed = sim.LEV_shed != -1
    sizes = 1 + np.arange(1, sim.nt) + np.cumsum(shed[1:])            # wake size at every step's roll-up
    pairs = float(np.sum((sizes + 80.0) * sizes))                     # ... x (wake + 80 bound vortices) sources
    rec = {"workload": "config 2: NACA0012 sinusoidal pitch, dt=1e-3, t in [0,50], full time_loop, fp32, device-resident march",
           "steps": int(sim.nt - 1), "wall_s": wall, "setup_s": t_setup, "time_loop_s": wall - t_setup, "final_wake": int(sizes[-1]),
           "rollup_pairs": pairs, "pairs_per_s_wall": pairs / wall, "unit": "pairs/s", "pair_kernel_s": kms * nl * 1e-3,
           "pair_kernel_launches": int(nl), "first_lev_step": int(np.argmax(shed)), "Cl_mean_last_period": float(np.mean(sim.Cl[-10000:]))}
    g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_config2_first1500.npz"))
    rec.update(first_lev_step_reference=int(g7["first_lev_step"]),
               max_abs_dCl_first_600_steps_vs_reference=float(np.abs(sim.Cl[:601] - g7["Cl"][:601]).max()), bound=1e-5,
               shedding_identical_through_step_1400=bool(np.array_equal(shed[:1401], g7["LEV_shed"][:1401] != -1)))
    del sim, eng
    return rec


def leg_config1(R):
    """BASELINE config 1 (the reference's README example, its own CPU-runnable case): wall time of LUDVM(**README_CASE) on the
    GPU path (precision 'auto' = float64 pair sums at this size), and `cpu_baseline_time_loop`: the same case by the oracle's
    restatement of LUDVM.py:597-1171 as the reference writes it (float64 NumPy, one core) -- BASELINE.md section 4's plan."""
    from ludvm_amd import LUDVM
    eng = R.rig.new_engine()
    LUDVM(**README_CASE, verbose=False, engine=eng)                   # (allocations, first launches)
    R.fence()
    t0 = time.perf_counter()
    sim = LUDVM(**README_CASE, verbose=False, engine=eng)
    R.fence()
    wall = time.perf_counter() - t0
    from oracle import ludvm_oracle as O          # the reported-only CPU baseline and the checker
    t0 = time.perf_counter()
    ref = O.OracleLUDVM(**README_CASE)
    cpu_s = time.perf_counter() - t0
    rec = {"workload": "config 1: README case, NACA0012, dt=5e-2, t in [0,20], Npoints=81, Ncoeffs=30, LESPcrit=0.2 (401 steps)",
           "wall_s": wall, "steps": int(sim.nt - 1), "precision": sim.precision,
           "cpu_baseline_time_loop": {"value": cpu_s, "unit": "s", "cores": 1, "kind": "port",
                                      "sample": "the whole case (401 steps) by oracle/ludvm_oracle.py::OracleLUDVM, the float64 NumPy "
                                                "restatement of LUDVM.py:597-1171 (pair sums as the reference writes them, :549-570; the "
                                                "reference's own file took 4.4 s in the build container, SURVEY section 6)"},
           "speedup_vs_cpu_baseline": cpu_s / wall,
           "lev_shedding_identical": bool(np.array_equal(sim.LEV_shed, ref.LEV_shed)),
           "max_abs_dCl_first_100_steps_vs_oracle": float(np.abs(sim.Cl[:101] - ref.Cl[:101]).max()), "bound": 1e-9}
    del sim, eng
    return rec


def other_configs(R, dx, dz, dg, repeats, cfg4_rec):
    """-> {"config5_flowfield": ..., "config2_time_loop": ..., "config1_readme": ...}; a leg that fails is reported, it does not
    take the line with it."""
    out = {}
    for key, on, fn in (("config5_flowfield", R.args.cfg5, lambda: leg_config5(R, dx, dz, dg)),
                        ("config2_time_loop", R.args.cfg2, lambda: leg_config2(R)),
                        ("config1_readme", R.args.cfg1, lambda: leg_config1(R))):
        if not on:
            continue
        R.publish(f"the {key} leg", repeats, dict(out), cfg4_rec)
        try:
            out[key] = fn()
        except Exception as e:       # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    return out

# ======================================================================================================================
# config 4: the sharded self-advection step
# ======================================================================================================================
def collective_sweep(R, issuers):
    """The collectives of the class-level sharding at its threshold sizes, on the stream the step uses: for every issuer
    (the library's communicator and / or torch.distributed) an int64 all-reduce of 0.5 ... 8 MB and an all-gather of
    8 B x n / G per rank; SWEEP_WARMUP + SWEEP_REPS repetitions each, every repetition between two events; the mean and the
    fastest repetition, each the MAX over ranks.  -> the record, with `min_wake_suggested`."""
    torch, dist, rig, eng, world = R.torch, R.dist, R.rig, R.eng, R.world
    mark, elapsed_ms = rig.stopwatch()
    sizes = sorted(SHARD_SAVED_US)                              # wake sizes whose step all-reduces 16 n bytes
    buf = torch.zeros([16 * sizes[-1] // 8], dtype=torch.int64, device=R.device)
    per_max = (max(SWEEP_GATHER_TARGETS) + world - 1) // world
    send = torch.zeros([2 * per_max], dtype=torch.float32, device=R.device)
    recv = torch.zeros([2 * per_max * world], dtype=torch.float32, device=R.device)

    def timed(fn):
        for _ in range(SWEEP_WARMUP):
            fn()
        R.fence()
        marks = [mark()]
        for _ in range(SWEEP_REPS):
            fn()
            marks.append(mark())
        rig.sync()
        reps = [elapsed_ms(a, b) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
        return {"mean_us": R.over_ranks(sum(reps) / len(reps)), "min_us": R.over_ranks(min(reps))}

    rec = {"reps": SWEEP_REPS, "warmup": SWEEP_WARMUP, "ranks": world,
           "what": "max over ranks of a rank's mean / fastest repetition, events on the launch stream"}
    for who in issuers:
        ar, ag = {}, {}
        for nv in sizes:
            cnt = 16 * nv // 8
            view = buf[:cnt]
            if who == "library":
                fn = lambda: eng.comm_allreduce_i64_dev(view.data_ptr(), cnt)                    # noqa: E731
            else:
                fn = lambda: dist.all_reduce(view, op=dist.ReduceOp.SUM)                         # noqa: E731
            ar[str(16 * nv)] = dict(timed(fn), wake_vortices=nv)
        for nt in SWEEP_GATHER_TARGETS:
            per = (nt + world - 1) // world
            s, r = send[: 2 * per], recv[: 2 * per * world]
            if who == "library":
                fn = lambda: eng.comm_allgather_dev(s.data_ptr(), r.data_ptr(), 8 * per)         # noqa: E731
            else:
                fn = lambda: dist.all_gather_into_tensor(r, s)                                   # noqa: E731
            ag[str(8 * per)] = dict(timed(fn), targets=nt)
        rec[who] = {"allreduce_i64_by_bytes": ar, "allgather_by_bytes_per_rank": ag}
    # the threshold that follows: the smallest tabulated wake whose all-reduce (the issuer the class would use: the library's
    # when there is one) costs less than half of what the split saves there; savings tabulated for 2, 4, 8 ranks -- the
    # largest of those not above this world (fewer ranks save less: conservative)
    g_tab = max([g for g in (2, 4, 8) if g <= world] or [2])
    who = issuers[0]
    suggested = None
    margins = {}
    for nv in sizes:
        cost = rec[who]["allreduce_i64_by_bytes"][str(16 * nv)]["mean_us"]
        saved = SHARD_SAVED_US[nv][g_tab]
        margins[str(nv)] = {"allreduce_us": cost, "saved_us": saved, "saved_over_cost": saved / cost if cost > 0 else None}
        if suggested is None and cost < 0.5 * saved:
            suggested = nv
    rec["min_wake_rule"] = (f"smallest tabulated wake whose {who} all-reduce (mean) < 1/2 of G{g_tab}_saved_us "
                            "(profiles/r04_shard_break_even.txt)" + ("; ONE rank: identities, not a measurement of xGMI" if world == 1 else ""))
    rec["min_wake_margins"] = margins
    return rec, suggested


def run_config4(R):
    args, rig, dist, world, rank, n = R.args, R.rig, R.dist, R.world, R.rank, R.n
    from ludvm_amd.sharded import ShardedWake
    symmetric, variant = R.symmetric, R.variant
    hard_exit = False
    coll_note = None
    coll = args.collectives if args.collectives != "auto" else ("library" if (R.backend == "nccl" and world > 1) else "torch")
    if coll == "library":
        R.rep.phase = "joining the library's RCCL communicator"
        coll, coll_note, abandoned = join_library_communicator(R.eng, rank, world, rig, R.backend, dist, args.comm_init_timeout)
        if abandoned:
            # a join that never returned still holds the old context: a fresh one for the rest of the run, and no
            # interpreter shutdown at the end (it would wait for the stuck call)
            R.eng = R.start_engine()
            hard_exit = True
    R.coll = coll
    eng = R.eng
    # a one-rank run issues its collectives all the same when it can (identities on the real RCCL): the library's
    # communicator always can, torch's needs the process group (LUDVM_BENCH_FORCE_DIST=1)
    force_coll = world == 1 and (coll == "library" or R.force_dist)

    def make_wake(sym):
        return ShardedWake(R.x, R.z, R.g, V_CORE, DT, rig.shard_kernel(eng), R.device, symmetric=sym, collectives=coll,
                           force_collectives=force_coll)

    def collective_words(sym):
        return ("one all_reduce(sum) of int64[2 N + 1] fixed-point sums per step" if sym
                else "one all_gather of fp32[2, N / G] positions per step") + \
            (" (ncclAllReduce / ncclAllGather issued inside libludvm_hip.so on its own communicator)" if coll == "library"
             else " (torch.distributed)")

    def run_variant(sym, steps, warmup):
        eng.set_symmetric(1 if sym else 0)      # (a one-rank "direct" block is the whole array: keep it on the direct kernel)
        wk = make_wake(sym)
        for _ in range(warmup):
            wk.step()
        R.fence()
        el, kms, nl, cms, nc = R.timed_region(wk.step, steps, wk)
        k_all, c_all = R.per_rank(kms), R.per_rank(cms)
        rec = {"kernel_variant": "symmetric" if sym else "direct", "collective": collective_words(sym),
               "value": wk.pairs_per_step * steps / el, "unit": "pairs/s", "steps": steps, "warmup": warmup,
               "ms_per_step": el / steps * 1e3,
               "pair_kernel_ms_per_rank": k_all, "pair_kernel_ms_max_over_mean": max(k_all) / (sum(k_all) / len(k_all)) if sum(k_all) > 0 else None,
               "collective_ms_per_rank": c_all, "collective_ms_max_over_mean": max(c_all) / (sum(c_all) / len(c_all)) if sum(c_all) > 0 else None,
               "collective_ms_min_over_ranks": min(c_all), "collectives_timed_per_rank": nc,
               "collective_bytes_per_rank": (16 * wk.n_pad + 8) if sym else 8 * wk.n_loc}
        return wk, rec, el, kms, nl

    R.rep.phase = f"warm-up and the reported timed region (config 4, {variant} variant)"
    wake, main_rec, elapsed, kernel_ms, launches = run_variant(symmetric, args.steps, args.warmup)
    desc = (f"config 4: synthetic wake N={n}, sharded over {world} GPU(s); per step: "
            + ("symmetric kernel on the rank's I-tile block of the unordered pairs + ONE all-reduce of the 64-bit "
               "fixed-point sums + replicated Euler update" if symmetric else
               "all-pairs kernel on own N/G targets + Euler update + ONE all-gather of positions"))
    R.M.update(elapsed=elapsed, kernel_ms=kernel_ms, launches=launches, pairs_per_step=wake.pairs_per_step,
               pairs_per_launch=(float(wake.n_pad) * float(wake.n_pad) / world if symmetric else float(wake.n_loc) * float(wake.n_pad)),
               desc=desc, collective=main_rec["collective"], coll_note=coll_note,
               per_rank_ms=main_rec["pair_kernel_ms_per_rank"], per_rank_coll_ms=main_rec["collective_ms_per_rank"],
               ns_l=wake.n_pad, nt_l=(wake.n_pad if symmetric else wake.n_loc))
    extra_main = {main_rec["kernel_variant"] + "_variant": dict(main_rec, reported_as_value=True)}
    R.publish("the other step variant", (), extra_main)

    # the other variant, then the repeats, within the budget; the checks (two steps + the oracle's ~2e9 pairs) keep a reserve
    step_s = elapsed / args.steps
    other_sym = not symmetric
    other_step_s = step_s * (0.7 if other_sym else 1.6)          # direct / symmetric ~ 1.5 [MI355X, one GPU]
    reserve = (step_s + other_step_s + 40.0) if args.check else 5.0
    wake_o = other_rec = None
    if args.other_variant and n >= 16384:
        fit = int((R.left() - reserve) * 0.8 / other_step_s) - 1            # (one warm-up step)
        o_steps = min(args.steps, fit)
        if o_steps >= 1:
            wake_o, other_rec, _, _, _ = run_variant(other_sym, o_steps, 1)
        else:
            other_rec = {"kernel_variant": "symmetric" if other_sym else "direct", "skipped":
                         f"budget: {R.left():.0f} s left of --budget-s {args.budget_s:.0f}, a step takes ~{other_step_s:.1f} s"}
    n_rep = max(0, min(args.repeats, int((R.left() - reserve) / max(elapsed, 1e-9))))
    extra = dict(extra_main)
    if other_rec is not None:
        extra[other_rec["kernel_variant"] + "_variant"] = dict(other_rec, reported_as_value=False)
    R.publish("the repeat regions (config 4)", (), extra)

    eng.set_symmetric(1 if symmetric else 0)
    repeats = []
    for _ in range(n_rep):
        repeats.append(R.timed_region(wake.step, args.steps, wake)[0])
        R.publish("the repeat regions (config 4)", repeats, extra)

    if args.check:
        checks = {}
        R.publish(f"the result check of the {main_rec['kernel_variant']} variant", repeats, extra)
        checks[main_rec["kernel_variant"]] = check_wake(wake, R.g, rank, rig)
        if wake_o is not None:
            R.publish(f"the result check of the {other_rec['kernel_variant']} variant", repeats, dict(extra, result_check=dict(checks)))
            eng.set_symmetric(1 if other_sym else 0)
            checks[other_rec["kernel_variant"]] = check_wake(wake_o, R.g, rank, rig)
        extra["result_check"] = checks

    # what the class-level sharding's collectives cost on this machine (needs collectives that are really issued)
    if args.sweep and (world > 1 or force_coll):
        R.publish("the collective micro-sweep", repeats, extra)
        del wake, wake_o
        issuers = [coll] + (["torch"] if (coll == "library" and (world > 1 or R.force_dist)) else [])
        try:
            sweep, suggested = collective_sweep(R, issuers)
            extra = dict(extra, collective_sweep_us=sweep, min_wake_suggested=suggested)
        except Exception as e:       # noqa: BLE001  (an optional phase: reported, never fatal)
            extra = dict(extra, collective_sweep_us={"error": f"{type(e).__name__}: {e}"}, min_wake_suggested=None)
        R.publish("writing the line", repeats, extra)
    return repeats, extra, None, hard_exit


def main(argv=None, rig_factory=HipRig):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    t_process = time.perf_counter()
    no_launcher = "RANK" not in os.environ and os.environ.get("WORLD_SIZE", "1") in ("", "1")     # (a bare WORLD_SIZE=1 some shells
    if args.gpus > 1 and no_launcher:                                                          #  export is no launcher either)
        # no launcher: be one (before torch is imported or the GPU touched in any way; a child process, never an exec)
        sys.exit(launch_ranks(args.gpus, argv, args.deadline_s))

    R = Run(args, rig_factory, t_process)
    # CPU baseline first (rank 0, one GPU): ~20 s of host work, then the GPU phase runs uninterrupted to the end
    if R.world == 1 and R.rank == 0 and args.cpu_rows > 0:
        R.cpu_rec, R.cpu_u, R.cpu_w = cpu_baseline(R.x, R.z, R.g, args.cpu_rows, args.cpu_budget)
    R.eng = R.start_engine()
    R.eng.set_symmetric(args.symmetric)
    R.info = R.eng.device_info()

    if R.world > 1:
        # RCCL builds its communicators on the first collective of each kind: do that outside the measurement
        # even when --warmup is 0
        torch, dist = R.torch, R.dist
        probe = torch.zeros(R.world * 4, dtype=torch.float32, device=R.device)
        piece = torch.ones(4, dtype=torch.float32, device=R.device)
        dist.all_gather_into_tensor(probe, piece)
        dist.all_reduce(torch.ones(4, dtype=torch.int64, device=R.device))     # the symmetric variant's collective
        R.rig.sync()

    hard_exit = False
    if R.workload == "cfg3":
        repeats, extra, cfg4_rec = run_config3(R)
    else:
        repeats, extra, cfg4_rec, hard_exit = run_config4(R)

    R.rep.phase = "writing the line"
    R.rep.emit(R.make_out(repeats, extra, cfg4_rec) if R.rank == 0 else None)
    # the deadline keeps watching the teardown: the line is out, a communicator that does not come down ends in exit 0
    R.rep.phase = "tearing down (the line has been written)"
    if hard_exit:
        sys.stderr.flush()
        os._exit(0)          # (a communicator join that never returned: do not wait for it at interpreter shutdown)
    if R.coll == "library":
        R.rig.sync()
        R.eng.comm_destroy()
    if R.world > 1 or R.force_dist:
        R.dist.barrier()
        R.dist.destroy_process_group()
    R.rep.finish()
    if R.world > 1:
        # the line is out and both communicators are down: leave without the interpreter's shutdown (library destructors at
        # exit are the one place left where a rank could hang unwatched, and a launcher waits for its slowest rank)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
