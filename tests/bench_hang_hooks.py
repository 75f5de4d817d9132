"""TEST-ONLY: misbehaviour injected on purpose into a run of bench.main(), through the two places where bench.py hands control
to its rig (HipRig.at_milestone / at_comm_join: no-ops on the machine).  bench.py itself reads no test variable (VERDICT r5
item 6); these hooks do, in the test rigs only (tests/bench_hang_rig.py: the HIP engine; tests/bench_cpu_rig.py: the oracle
under gloo):

    LUDVM_BENCH_TEST_HANG=1            the first milestone never ends (a late, optional phase that hangs)
    LUDVM_BENCH_TEST_HANG_COMM=1       the join of the library's communicator never returns
    LUDVM_BENCH_TEST_HANG_COMM=late    ... returns after the timeout (a join that was merely slow)
    LUDVM_BENCH_TEST_HANG_COMM=raise   the known-sum proof on the joined communicator raises (an RCCL error on the first collective)
"""
import os
import time


class HangHooks:
    def at_milestone(self, reporter, phase):
        if os.environ.get("LUDVM_BENCH_TEST_HANG") == "1":
            reporter.phase = phase + " [test hook: hung on purpose]"
            while True:
                time.sleep(1.0)

    def at_comm_join(self, stage, timeout_s):
        hook = os.environ.get("LUDVM_BENCH_TEST_HANG_COMM", "")
        while stage == "before" and hook == "1":
            time.sleep(1.0)
        if stage == "after" and hook == "late":
            time.sleep(timeout_s + 2.0)
        if stage == "proof" and hook == "raise":
            raise RuntimeError("ludvm_comm_allreduce_i64_dev: RCCL error [test hook: raised on purpose]")
