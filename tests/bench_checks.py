"""What a config-4 line of bench.py must carry; shared by the GPU tier (tests/test_gpu_bench.py) and the CPU rehearsal of
the N > 1 control flow (tests/test_bench_cpu_rehearsal.py)."""
import pytest


def assert_self_checking_config4(d, ranks, main="symmetric", collectives_issued=True, min_value=1e10):
    """What every config-4 line must carry: the collective's own time per rank beside the pair
    kernel's, BOTH step variants, and a result check that passed."""
    c = d["config"]
    assert len(c["pair_kernel_ms_per_rank"]) == ranks and len(c["collective_ms_per_rank"]) == ranks
    assert c["pair_kernel_ms_max_over_mean"] >= 1.0
    assert c["class_sharding_thresholds"] == {"min_wake": 131072, "min_targets": 65536, "min_pairs": 2**30}
    other = "direct" if main == "symmetric" else "symmetric"
    for name, reported in ((main, True), (other, False)):
        v = d[name + "_variant"]
        assert v["kernel_variant"] == name and v["reported_as_value"] is reported and "skipped" not in v, v
        assert v["value"] > min_value and v["steps"] >= 1 and v["ms_per_step"] > 0
        assert len(v["pair_kernel_ms_per_rank"]) == ranks and all(t > 0 for t in v["pair_kernel_ms_per_rank"])
        assert len(v["collective_ms_per_rank"]) == ranks
        if collectives_issued:
            assert all(t > 0 for t in v["collective_ms_per_rank"]) and v["collectives_timed_per_rank"] == v["steps"]
            assert v["collective_ms_max_over_mean"] >= 1.0
        # the kernel and the collective are both inside the step
        assert max(v["pair_kernel_ms_per_rank"]) + min(v["collective_ms_per_rank"]) <= v["ms_per_step"] * 1.05
        assert ("all_reduce" if name == "symmetric" else "all_gather") in v["collective"]
    assert d[main + "_variant"]["value"] == pytest.approx(d["value"]) and d[main + "_variant"]["steps"] == d["steps"]
    assert c["collective_ms_per_rank"] == d[main + "_variant"]["collective_ms_per_rank"]
    for name in (main, other):
        k = d["result_check"][name]
        assert "error" not in k, k
        assert k["ranks_agree"] is True and k["finite"] is True and k["samples"] == 256
        assert k["gpu_vs_oracle_max_rel_err"] < 1e-5, k
    assert "config4_one_gpu.value of the N = 1 line" in d["scaling_denominator"]


def assert_sweep(d, ranks, issuers):
    """`collective_sweep_us` / `min_wake_suggested`: the class-level sharding's collectives at its
    threshold sizes, timed on this machine by the N > 1 line itself."""
    s = d["collective_sweep_us"]
    assert "error" not in s, s
    assert s["ranks"] == ranks and s["reps"] == 20 and s["warmup"] == 3
    for who in issuers:
        ar, ag = s[who]["allreduce_i64_by_bytes"], s[who]["allgather_by_bytes_per_rank"]
        assert sorted(int(k) for k in ar) == [524288, 1048576, 2097152, 4194304, 8388608]
        assert [v["wake_vortices"] for v in ar.values()] == [32768, 65536, 131072, 262144, 524288]
        assert [v["targets"] for v in ag.values()] == [65536, 262144, 1000000]
        assert [int(k) for k in ag] == [8 * ((t + ranks - 1) // ranks) for t in (65536, 262144, 1000000)]
        for v in list(ar.values()) + list(ag.values()):
            assert 0 < v["min_us"] <= v["mean_us"]
    m = s["min_wake_margins"]
    assert sorted(int(k) for k in m) == [32768, 65536, 131072, 262144, 524288]
    sug = d["min_wake_suggested"]
    assert sug is None or sug in (32768, 65536, 131072, 262144, 524288)
    for k, v in m.items():            # the rule, recomputed: the first size with cost < saved / 2
        if sug is not None and int(k) < sug:
            assert v["allreduce_us"] >= 0.5 * v["saved_us"]
    if sug is not None:
        assert m[str(sug)]["allreduce_us"] < 0.5 * m[str(sug)]["saved_us"]
