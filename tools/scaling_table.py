#!/usr/bin/env python3
"""Strong-scaling table from bench.py's lines at N = 1, 2, 4, 8 (no GPU needed).

    python tools/scaling_table.py line_n1.json line_n2.json ...      # files holding a bench line, or a driver record with "parsed"

The N = 1 line's `value` is config 3 (N = 1e6) and the N > 1 lines' `value` is config 4 (N = 8e6): the same-work denominator of
"G GPUs vs 1" is the N = 1 line's `config4_one_gpu.value` (bench.py, `scaling_denominator`).  Per line: pairs/s, ms per step,
speed-up and efficiency against that denominator, the slowest rank's pair kernel and collective per step, both step variants, the
result checks, and -- from the first line that carries it -- what the run says about the class-level sharding's thresholds."""
import json
import sys


def load(path):
    with open(path) as f:
        text = f.read()
    try:
        d = json.loads(text)
    except ValueError:
        d = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
    if isinstance(d, dict) and "parsed" in d and isinstance(d["parsed"], dict):
        d = d["parsed"]
    if isinstance(d, dict) and "runs" in d:          # a record holding several lines
        return [r.get("parsed", r) for r in d["runs"]]
    return [d]


def table(lines):
    lines = sorted((l for l in lines if isinstance(l, dict) and "value" in l), key=lambda l: l["n_gpus"])
    one = next((l for l in lines if l["n_gpus"] == 1), None)
    base = (one or {}).get("config4_one_gpu", {}).get("value")
    rows = []
    for l in lines:
        g = l["n_gpus"]
        cfg4 = "config 4" in l["config"]["workload"]
        c = l["config"]
        row = {"n_gpus": g, "workload": "config 4" if cfg4 else "config 3", "value": l["value"], "ms_per_step": l["ms_per_step"],
               "incomplete": l.get("incomplete")}
        if cfg4 and base:
            row["speedup_vs_config4_one_gpu"] = l["value"] / base
            row["efficiency"] = l["value"] / base / g
        if cfg4:
            row["pair_kernel_ms_max"] = max(c["pair_kernel_ms_per_rank"])
            row["collective_ms_max"] = max(c["collective_ms_per_rank"]) if c.get("collective_ms_per_rank") else None
            for v in ("symmetric", "direct"):
                rec = l.get(v + "_variant")
                if rec and "value" in rec:
                    row[v + "_value"] = rec["value"]
                chk = (l.get("result_check") or {}).get(v)
                if chk:
                    row[v + "_check"] = {"ranks_agree": chk.get("ranks_agree"), "gpu_vs_oracle_max_rel_err": chk.get("gpu_vs_oracle_max_rel_err")}
            if "min_wake_suggested" in l:
                row["min_wake_suggested"] = l["min_wake_suggested"]
        elif g == 1:
            row["config4_one_gpu_value"] = base
        rows.append(row)
    return rows


def main():
    lines = [l for p in sys.argv[1:] for l in load(p)]
    rows = table(lines)
    print("| GPUs | workload | pairs/s | ms / step | speed-up vs config4_one_gpu | efficiency | pair kernel ms (slowest rank) | "
          "collective ms (slowest rank) | direct variant pairs/s | checks | min_wake_suggested |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    f = lambda v, fmt: "" if v is None else format(v, fmt)          # noqa: E731
    for r in rows:
        checks = "; ".join(f"{v}: agree={r[v + '_check']['ranks_agree']}, err={r[v + '_check']['gpu_vs_oracle_max_rel_err']:.1e}"
                           for v in ("symmetric", "direct") if r.get(v + "_check") and r[v + "_check"]["gpu_vs_oracle_max_rel_err"] is not None)
        print(f"| {r['n_gpus']} | {r['workload']}{' (INCOMPLETE)' if r['incomplete'] else ''} | {r['value']:.4g} | {r['ms_per_step']:.2f} | "
              f"{f(r.get('speedup_vs_config4_one_gpu'), '.2f')} | {f(r.get('efficiency'), '.3f')} | {f(r.get('pair_kernel_ms_max'), '.2f')} | "
              f"{f(r.get('collective_ms_max'), '.3f')} | {f(r.get('direct_value'), '.4g')} | {checks} | {r.get('min_wake_suggested', '')} |")
    if rows and rows[0]["n_gpus"] == 1 and rows[0].get("config4_one_gpu_value"):
        print(f"\nsame-work denominator (config 4 on ONE GPU, from the N = 1 line): {rows[0]['config4_one_gpu_value']:.4g} pairs/s")
    return rows


if __name__ == "__main__":
    main()
