#!/usr/bin/env python3
"""TEST-ONLY: bench.py on the real machine (HipRig: the HIP engine on cuda:<local_rank>) with the misbehaviour of
tests/bench_hang_hooks.py injected on purpose.

    LUDVM_BENCH_TEST_HANG=1 python tests/bench_hang_rig.py --gpus 1 ...        (same arguments as bench.py)

Everything that runs is bench.py's own code -- bench.main() with a rig that differs from HipRig in the two hook methods
only; self-launched ranks (`--gpus N` without a launcher) run THIS script, sys.argv[0].  Used by tests/test_gpu_bench.py for
the deadline and the communicator-join watchdog."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ == "__main__":
    import bench
    from bench_hang_hooks import HangHooks

    class HangHipRig(HangHooks, bench.HipRig):
        pass
    bench.main(rig_factory=HangHipRig)
