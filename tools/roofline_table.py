#!/usr/bin/env python3
"""Per-config roofline table of DESIGN.md section 4, recomputed from the rocprofv3 summaries under profiles/ (no GPU needed).

    python tools/roofline_table.py [round_prefix]            # default r06

Per kernel: pairs per launch (stated in DESIGN.md section 4), average duration from `*_kernel_stats.csv`, credited fraction =
13 FLOP x pairs / ns / 157.3e12, issued fraction (the FLOP per ordered pair the kernel really executes), VALU
wave-instructions per launch from `*_pmc_sq.csv` against the instruction model, held clock = GRBM_GUI_ACTIVE / 8 XCDs /
duration, time against the issue model at that clock, LDS bank-conflict cycles, and HBM-side traffic (2 x FETCH_SIZE +
WRITE_SIZE, KB as rocprofv3 reports them; MI355X_MICROARCH.md's gfx950 correction) over the algorithmic bytes."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 157.3e12
SIMDS = 256 * 4

# (row label, file prefix, kernel-name substring, ordered pairs per launch, executed FLOP per ordered pair,
#  VALU wave-instructions per 64-lane "pair-lane" group: instructions per ordered pair and lane,
#  issue cycles per ordered pair and lane (packed op 4, v_rsq_f32 8; measured in tools/ubench), algorithmic HBM bytes)
N3, G5 = 1_000_000, 4096 * 4096
ROWS = [
    ("config 3 / 4: symmetric, quad variant", "bench_cfg3_sym", "pair_sym_quad_f32<8>", float(N3) ** 2, 9,
     13 / 4, (11 * 4 + 2 * 8) / 4, 28.0 * N3),
    ("config 3 direct (`--symmetric 0`; every non-self call)", "bench_cfg3_direct", "pair_f32<2, 1024, false, 0, false>", float(N3) ** 2, 13,
     10 / 2, (8 * 4 + 2 * 8) / 2, 28.0 * N3),
    ("config 5: 4096^2 grid x 1e6 sources, 4 x 4 patch", "config5", "pair_f32<16, 1024, false, 2, false>", float(G5) * N3, 10.5 * 2 / 2 + 0,
     7.25 / 2, (5.25 * 4 + 2 * 8) / 2, 12.0 * N3 + 8.0 * G5),
    ("config 2 at ~60 000 vortices: symmetric, mixed granularity", "config2_sizes", "pair_sym_f32<8, false, 0, true>", 60075.0 ** 2, 9,
     13 / 4, (11 * 4 + 2 * 8) / 4, 28.0 * 60075),
    ("config 2 at ~36 000 vortices: symmetric, 4 waves per item", "config2_sizes", "pair_sym_f32<8, false, 4, true>", 36075.0 ** 2, 9,
     13 / 4, (11 * 4 + 2 * 8) / 4, 28.0 * 36075),
]


T8_SWITCH = {"r05": 34816}


def stats(prefix):
    out = {}
    with open(prefix + "_kernel_stats.csv") as f:
        for r in csv.DictReader(f):
            out[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    return out


def pmc(prefix, which):
    out = {}
    with open(f"{prefix}_pmc_{which}.csv") as f:
        for r in csv.DictReader(f):
            out[(r["kernel"], r["counter"])] = float(r["mean_per_dispatch"])
    return out


def config2_rows(rnd):
    """Config 2's rows from config 2's OWN run (round 5): `{rnd}_config2_{kernel_stats,pmc_*}.csv` are
    rocprofv3 passes of the full 50 000-step time loop (tools/run_configs.py cfg2) and `{rnd}_config2_under_rocprof.out` its JSON
    line with the wake size after every 250th step.  The wake only grows and the launch geometry is a function of the wake size
    alone, so which variant of the symmetric kernel served a step -- direct (small wakes), <4,.,4>, <4,.,0> (mixed granularity),
    <8,.,4>, <8,.,0> -- follows from the step's wake size by the library's rule; pairs per launch = the mean of n^2 over a
    variant's steps.  Rounds without these files (r04) fall back to the proxy rows in ROWS."""
    import json
    import re
    p = os.path.join(ROOT, "profiles", f"{rnd}_config2")
    if not os.path.exists(p + "_pmc_sq.csv"):
        return None
    with open(p + "_under_rocprof.out") as f:
        run = json.loads([l for l in f if l.startswith("{")][-1])
    every = run["wake_size_every_250_steps"]
    step_of = [1 + 250 * k for k in range(len(every))]                  # sizes[k] is the wake after step k + 1

    def wake_at(step):
        import bisect
        k = min(len(every) - 2, max(0, bisect.bisect_right(step_of, step) - 1))
        t = (step - step_of[k]) / 250.0
        return every[k] + t * (every[k + 1] - every[k])
    st = stats(p)
    # Which kernel served which step: the library's own rule (launch.hip: sym_tile_t; pair_sym_kernels.hpp: sym_geometry_t),
    # restated here, applied to the wake size of every step.  Round 6: rounds 3-5 assumed that each variant serves ONE contiguous
    # range of steps; the waves-per-item rule follows the tile count's parity where the d-chunks stop dividing evenly, so
    # <4,.,4> comes back between ~32 900 and the T = 8 switch and <8,.,4> for the last ~900 steps -- the old attribution put
    # those launches' pairs into the wrong rows (the call counts below are checked against the statistics file instead).
    t8 = T8_SWITCH.get(rnd, 36864)

    def variant(n):
        if n < 11264:
            return "pair_f32<1, 256, false, 0, true>"
        T = 8 if n >= t8 else 4
        W = 64 * T
        nt = max(1, (n + W - 1) // W)
        dtot = (nt - 1) // 2 + (1 if (nt % 2 == 0 and nt > 1) else 0)
        ys = max(1, min(64, dtot))
        if dtot > 0:
            per0 = (dtot + ys - 1) // ys
            ys = (dtot + per0 - 1) // per0
        rs = 1
        while rs < 4 and nt * ys * rs < 10500:
            rs *= 2
        return f"pair_sym_f32<{T}, false, {0 if rs < 4 else 4}, true>"
    served = {}
    for s_ in range(1, run["steps"] + 1):
        served.setdefault(variant(int(wake_at(s_))), []).append(s_)
    rows = []
    for short in ("pair_sym_f32<4, false, 4, true>", "pair_sym_f32<4, false, 0, true>", "pair_sym_f32<8, false, 4, true>",
                  "pair_sym_f32<8, false, 0, true>"):
        steps = served.get(short, [])
        calls = [c for name, (c, _) in st.items() if short + "(" in name]
        assert steps and calls and abs(len(steps) - calls[0]) <= 0.03 * calls[0] + 5, (short, len(steps), calls)
        n2 = sum(wake_at(q) ** 2 for q in steps) / len(steps)
        ranges, lo = [], steps[0]
        for q0, q1 in zip(steps, steps[1:] + [None]):
            if q1 != q0 + 1:
                ranges.append(f"{wake_at(lo):.0f}-{wake_at(q0):.0f}")
                lo = q1
        rows.append((f"config 2, {calls[0]} steps: wakes of {' and '.join(ranges)} vortices", "config2", short, n2, 9,
                     13 / 4, (11 * 4 + 2 * 8) / 4, 28.0 * n2 ** 0.5))
    direct = [c for name, (c, _) in st.items() if "pair_f32<1, 256, false, 0, true>(" in name]
    assert direct and abs(len(served["pair_f32<1, 256, false, 0, true>"]) - direct[0]) <= 5
    return rows


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
    own = config2_rows(rnd)
    table = [r for r in ROWS if r[1] != "config2_sizes"] + own if own else ROWS
    print("| kernel (workload) | pairs / launch | avg ms | credited frac (13 FLOP) | issued frac | VALU insts / launch (vs model) | "
          "held clock GHz | time vs issue model | LDS conflict cycles | HBM traffic / algorithmic bytes | source (profiles/) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for label, pre, kern, pairs, exe, ipp, cpp, alg in table:
        p = os.path.join(ROOT, "profiles", f"{rnd}_{pre}")
        ks = [(k, v) for k, v in stats(p).items() if kern + "(" in k]
        assert len(ks) == 1, (kern, ks)
        calls, ns = ks[0][1]
        sq, fe, wr = pmc(p, "sq"), pmc(p, "fetch"), pmc(p, "write")
        key = [k for k in sq if k[0].endswith(kern)][0][0]
        valu, gui, lds = sq[(key, "SQ_INSTS_VALU")], sq[(key, "GRBM_GUI_ACTIVE")], sq[(key, "SQ_LDS_BANK_CONFLICT")]
        fetch_kb, write_kb = fe[(key, "FETCH_SIZE")], wr[(key, "WRITE_SIZE")]
        frac = 13 * pairs / (ns * 1e-9) / PEAK
        model_insts = pairs / 64 * ipp
        clock = gui / 8 / (ns * 1e-9)
        model_s = pairs / 64 * cpp / SIMDS / clock
        traffic = (2 * fetch_kb + write_kb) * 1024
        print(f"| `{kern}` ({label}) | {pairs:.4g} | {ns * 1e-6:.3f} ({calls} launches) | **{frac:.3f}** | {frac * exe / 13:.3f} | "
              f"{valu:.4g} ({valu / model_insts:.3f} x) | {clock * 1e-9:.2f} | {model_s / (ns * 1e-9):.3f} | {lds:.3g} | "
              f"{traffic / 1e6:.0f} MB / {alg / 1e6:.1f} MB = {traffic / alg:.1f} x | `{rnd}_{pre}_*` |")


if __name__ == "__main__":
    main()
