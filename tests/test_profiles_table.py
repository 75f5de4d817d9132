"""CPU tier: DESIGN.md section 5's per-kernel roofline table is recomputable from the committed rocprofv3 summaries
(tools/roofline_table.py over profiles/r04_*), and the figures the docs quote follow from those files."""
import os
import subprocess
import sys

from conftest import ROOT


def test_roofline_table_recomputes_from_profiles():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), "r04"], capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [l for l in p.stdout.splitlines() if l.startswith("| `")]
    assert len(rows) == 5, p.stdout
    cells = [[c.strip() for c in r.strip("|").split("|")] for r in rows]
    by = {c[0].split("`")[1]: c for c in cells}
    quad = by["pair_sym_quad_f32<8>"]
    direct = by["pair_f32<2, 1024, false, 0, false>"]
    patch = by["pair_f32<16, 1024, false, 2, false>"]
    frac = lambda c: float(c[3].strip("*"))      # noqa: E731
    # credited fractions: the headline clears the north star's 40 % on every kernel; issued <= credited
    assert 0.70 < frac(quad) < 0.80 and 0.45 < frac(direct) < 0.54 and 0.62 < frac(patch) < 0.703
    for c in cells:
        assert float(c[4]) <= frac(c) + 1e-9 and frac(c) > 0.40, c
        assert 0.6 < float(c[7]) <= 1.0, c                 # measured time never beats the issue model
        assert c[8] == "0", c                              # no LDS bank conflict in any pair kernel (round 4)
        assert 1.0 <= float(c[5].split("(")[1].split("x")[0]) < 1.2, c      # VALU instructions within 20 % of the model
    # the direct kernel's distance from its 54 % ceiling is the held clock: >= 95 % of the issue model
    assert float(direct[7]) > 0.95


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
