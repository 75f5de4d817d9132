"""CPU tier: DESIGN.md section 5's per-kernel roofline table is recomputable from the committed rocprofv3 summaries
(tools/roofline_table.py over profiles/r06_*; r05_* / r04_* for the rounds before), and the figures the docs quote follow from
those files."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _table(rnd):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), rnd], capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [l for l in p.stdout.splitlines() if l.startswith("| `")]
    return [[c.strip() for c in r.strip("|").split("|")] for r in rows], p.stdout


@pytest.mark.parametrize("rnd,t8_switch", [("r06", 36864), ("r05", 34816)])
def test_roofline_table_recomputes_from_profiles(rnd, t8_switch):
    """The current table (round 6: after the T = 8 switch moved and the chain's waves got their priority) and round 5's: configs
    3 / 4 / 5, and config 2's rows from config 2's OWN full run (no proxy) -- one row per variant of the
    symmetric kernel the 50 000 steps went through."""
    cells, out = _table(rnd)
    assert len(cells) == 7, out
    by = {c[0].split("`")[1]: c for c in cells}
    quad = by["pair_sym_quad_f32<8>"]
    direct = by["pair_f32<2, 1024, false, 0, false>"]
    patch = by["pair_f32<16, 1024, false, 2, false>"]
    frac = lambda c: float(c[3].strip("*"))      # noqa: E731
    # credited fractions: the headline clears the north star's 40 % on every kernel; issued <= credited
    assert 0.70 < frac(quad) < 0.80 and 0.45 < frac(direct) < 0.54 and 0.62 < frac(patch) < 0.703
    for c in cells:
        assert float(c[4]) <= frac(c) + 1e-9 and frac(c) > 0.40, c
        assert 0.5 < float(c[7]) <= 1.0, c                 # measured time never beats the issue model
        assert c[8] == "0", c                              # no LDS bank conflict in any pair kernel
        assert c[10].startswith(f"`{rnd}_"), c             # every row names the files (and thereby the box) it comes from
    for c in (quad, direct, patch):
        assert 1.0 <= float(c[5].split("(")[1].split("x")[0]) < 1.2, c      # VALU instructions within 20 % of the model
    # the direct kernel's distance from its 54 % ceiling is the held clock: >= 95 % of the issue model
    assert float(direct[7]) > 0.95
    # config 2: the rows ARE the run -- their kernels are the symmetric kernels of the full run's statistics file, their
    # launches are that file's call counts, and the steps the library's rule (restated in tools/roofline_table.py) gives each
    # variant add up to the run (round 6: a variant may serve more than one range of wake sizes -- the waves-per-item rule
    # follows the parity of the tile count -- so the rows name their ranges instead of assuming one each)
    import csv
    import re
    cfg2 = [c for c in cells if "config 2, " in c[0]]
    assert [c[0].split("`")[1] for c in cfg2] == ["pair_sym_f32<4, false, 4, true>", "pair_sym_f32<4, false, 0, true>",
                                                  "pair_sym_f32<8, false, 4, true>", "pair_sym_f32<8, false, 0, true>"]
    with open(os.path.join(ROOT, "profiles", f"{rnd}_config2_kernel_stats.csv")) as f:
        stats = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"])) for r in csv.DictReader(f)}
    sym = {k: v for k, v in stats.items() if "pair_sym_f32<" in k}
    assert len(sym) == 4 and sum(v[2] for v in sym.values()) > 60.0          # > 60 % of the run's kernel time
    direct_calls = next(v[0] for k, v in stats.items() if "pair_f32<1, 256, false, 0, true>(" in k)
    total = direct_calls
    for c in cfg2:
        name = c[0].split("`")[1]
        calls, avg_ns, _ = next(v for k, v in sym.items() if name + "(" in k)
        assert int(re.search(r"config 2, (\d+) steps", c[0]).group(1)) == calls
        assert abs(float(c[2].split()[0]) - avg_ns * 1e-6) < 1e-3            # the row's duration IS the full run's
        assert 0.55 < frac(c) < 0.75, c                                      # mid sizes: between 0.58 and 0.71 credited
        total += calls
    assert total == 50000
    # the T = 8 tile takes over at 36 864 vortices (launch.hip, kSymT8MinN; 34 816 through round 5), whatever the proxy once showed
    assert abs(int(re.search(r"wakes of (\d+)-", cfg2[2][0]).group(1)) - t8_switch) < 1500
    # ... and the four-waves-per-item kernel comes back below the switch (the tile count's parity): two ranges
    assert " and " in cfg2[0][0] and " and " in cfg2[2][0] and " and " not in cfg2[1][0]


def test_round_4_table_still_recomputes():
    cells, out = _table("r04")
    assert len(cells) == 5, out


def test_bench_break_even_table_is_the_committed_measurement():
    """bench.py's SHARD_SAVED_US (what `min_wake_suggested` is computed from) is a copy of profiles/r04_shard_break_even.txt."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    rows = {}
    with open(os.path.join(ROOT, "profiles", "r04_shard_break_even.txt")) as f:
        for line in f:
            if line.startswith("{"):
                r = json.loads(line)
                rows[r["n"]] = r
    assert set(bench.SHARD_SAVED_US) <= set(rows)
    for n, by_g in bench.SHARD_SAVED_US.items():
        assert rows[n]["allreduce_bytes"] == 16 * n
        for g, us in by_g.items():
            assert rows[n][f"G{g}_saved_us"] == us, (n, g)


def test_bench_traffic_constants_are_the_committed_counter_passes():
    """bench.py's roofline.traffic (HBM-side bytes per launch of the dominant kernel at config 3) is a constant looked up by
    kernel: 2 x FETCH_SIZE + WRITE_SIZE (KB) of the counter passes the entry names, summed over the kernels of one launch."""
    import csv
    sys.path.insert(0, ROOT)
    import bench

    def kb(prefix, which, counter, kernels):
        with open(os.path.join(ROOT, f"{prefix}_pmc_{which}.csv")) as f:
            return sum(float(r["mean_per_dispatch"]) for r in csv.DictReader(f)
                       if r["counter"] == counter and any(k in r["kernel"] for k in kernels))
    for name, rec in bench.PMC_TRAFFIC_CFG3.items():
        prefix = rec["source"].split("_pmc_")[0]
        kernels = ("pair_sym_quad_f32<8>", "pair_sym_f32<8, false, 1, false>") if "quad" in name else ("pair_f32<2, 1024, false, 0, false>",)
        want = (2 * kb(prefix, "fetch", "FETCH_SIZE", kernels) + kb(prefix, "write", "WRITE_SIZE", kernels)) * 1024
        assert abs(rec["bytes"] - want) / want < 1e-4, (name, rec["bytes"], want)


def test_scaling_table_divides_by_the_same_work_figure():
    """tools/scaling_table.py on committed lines (the N = 1 line of the profiled box and the two-gloo-rank rehearsal on one card):
    the speed-up of an N > 1 line is taken against the N = 1 line's config4_one_gpu.value, not against its `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import json
    import scaling_table
    one = scaling_table.load(os.path.join(ROOT, "profiles", "r05_bench_cfg3_sym_bench.json"))
    two = scaling_table.load(os.path.join(ROOT, "profiles", "r05_bench_two_gloo_ranks_full_size.json"))
    rows = scaling_table.table(one + two)
    assert [r["n_gpus"] for r in rows] == [1, 2] and rows[0]["workload"] == "config 3" and rows[1]["workload"] == "config 4"
    base = one[0]["config4_one_gpu"]["value"]
    assert rows[1]["speedup_vs_config4_one_gpu"] == two[0]["value"] / base != two[0]["value"] / one[0]["value"]
    assert abs(rows[1]["efficiency"] - 0.5) < 0.05          # two ranks SHARING one card: the rehearsal, as it must, scales by 1.0
    assert rows[1]["symmetric_check"]["ranks_agree"] is True and rows[1]["min_wake_suggested"] == two[0]["min_wake_suggested"]
    assert "config4_one_gpu.value" in two[0]["scaling_denominator"]


def test_the_tables_rule_is_the_librarys_rule(tmp_path):
    """DESIGN section 5's config-2 rows and profiles/r06_mid_size_variant_table.txt attribute steps / mark picks by a Python
    restatement of the library's launch rule (tools/roofline_table.py::config2_rows, tools/r06_mid_size_sweep.py::rule_pick).
    The rule itself is host-callable C++ (pair_sym_kernels.hpp: sym_geometry; launch.hip: kSymT8MinN): a small host program
    built from the library's own header prints it for a ladder of wake sizes, and the restatements must agree everywhere."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "ludvm_amd", "csrc")
    t8 = int(re.search(r"constexpr long long kSymT8MinN = (\d+);", open(os.path.join(csrc, "launch.hip")).read()).group(1))
    src = tmp_path / "rule.hip"
    src.write_text('#include <cstdio>\n#include "pair_sym_kernels.hpp"\nint main() {\n'
                   f'  for (long long n = 11264; n <= 70000; n += 97) {{\n    const int T = n >= {t8} ? 8 : 4;\n'
                   '    const ludvm::SymGeom g = ludvm::sym_geometry(n, T, 0, 0);\n'
                   '    std::printf("%lld %d %d\\n", n, T, g.rsplit);\n  }\n  return 0;\n}\n')
    exe = tmp_path / "rule"
    p = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", f"-I{csrc}", "-o", str(exe), str(src)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lib = {int(a): (int(b), int(c)) for a, b, c in (l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True,
                                                                                      timeout=60).stdout.splitlines())}
    assert len(lib) > 600
    # restatement 1: the sweep's rule_pick (cut out of the tool: importing it would need an engine)
    text = open(os.path.join(ROOT, "tools", "r06_mid_size_sweep.py")).read()
    ns = {"os": os}
    exec(text[text.index("K_TARGET_WAVES, K_MAX_SPLIT"):text.index("CANDS =")], ns)
    assert ns["K_T8_MIN_N"] == t8
    # restatement 2: roofline_table's variant(), same cut
    text2 = open(os.path.join(ROOT, "tools", "roofline_table.py")).read()
    body = text2[text2.index("    def variant(n):"):text2.index("    served = {}")]
    ns2 = {"t8": t8}
    exec("def make(t8):\n" + body + "    return variant\n", ns2)
    variant = ns2["make"](t8)
    for n, (T, rs) in lib.items():
        # rsplit: 0 = mixed granularity (the size rule left room below four waves per item), 4 = four waves per item
        assert rs in (0, 4), (n, rs)
        assert ns["rule_pick"](n) == (T, "mixed" if rs == 0 else "x4"), n
        assert variant(n) == f"pair_sym_f32<{T}, false, {rs}, true>", n
