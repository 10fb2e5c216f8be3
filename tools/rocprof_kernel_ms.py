#!/usr/bin/env python3
"""profiles/rocprof_kernel_ms.json (what bench.py reports as roofline.kernel_ms_rocprof) from a tools/r03_final_measure.sh run.
usage: python tools/rocprof_kernel_ms.py <tag>        e.g. r05_final  (reads profiles/<tag>_*kernel_stats.csv and, for the per-rank share,
the kernel trace under gpurun_out/prof_<tag>_share8: its launches are of two kinds -- the 20 single-MSM latency runs and the grouped
8-MSM launches of the timed region -- so the --stats average over both says nothing; only the grouped ones are averaged here and the
per-launch list is written next to it as profiles/<tag>_share8_smvp_launches.csv)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats_avg(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "k_smvp_chunks" in r["Name"]:
                return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    raise SystemExit("no k_smvp_chunks in " + path)


def main():
    tag = sys.argv[1]
    prof = os.path.join(ROOT, "profiles")
    out = {}
    ms, calls = stats_avg(os.path.join(prof, tag + "_kernel_stats.csv"))
    out["logn20_endomorphism_single"] = {
        "k_smvp_chunks_avg_ms": ms, "calls": calls,
        "source": "profiles/%s_kernel_stats.csv: rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 1 --no-cpu-baseline` (tools/profile_round.sh %s)" % (tag, tag)}
    ms, calls = stats_avg(os.path.join(prof, tag + "_logn24_kernel_stats.csv"))
    out["logn24_endomorphism_single"] = {"k_smvp_chunks_avg_ms": ms, "calls": calls,
                                         "source": "profiles/%s_logn24_kernel_stats.csv: the same passes with --logn 24" % tag}
    # one rank's share of an 8-rank run: plain shares (2 of the 16 windows per MSM) and wide-table shares (1 of the 8 virtual windows)
    for suffix, key, what, env in (("share8", "logn20_plain_w2", "8 MSMs x 2 windows", "BENCH_EMULATE_WORLD=8"),
                                   ("share8_wide", "logn20_tables_wide_w1", "8 MSMs x 1 virtual window", "BENCH_EMULATE_WORLD=8 BENCH_BASES=tables_wide")):
        traces = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, suffix), "**", "*kernel_trace.csv"), recursive=True)
        if not traces:
            continue
        rows = []
        with open(traces[0]) as f:
            for r in csv.DictReader(f):
                if "k_smvp_chunks" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        rows.sort()
        lst = os.path.join(prof, "%s_%s_smvp_launches.csv" % (tag, suffix))
        with open(lst, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["launch", "grid_x_lanes_per_window", "grid_y_windows", "duration_us"])
            for i, r in enumerate(rows):
                w.writerow([i, r[1], r[2], "%.3f" % r[3]])
        wmax = max(r[2] for r in rows)
        grouped = [r[3] for r in rows if r[2] == wmax]
        out[key] = {
            "k_smvp_chunks_avg_ms": sum(grouped) / len(grouped) / 1e3, "calls": len(grouped),
            "source": "profiles/%s_%s_smvp_launches.csv: rocprofv3 --kernel-trace of `%s bench.py --steps 16 --warmup 8` (one rank's share of an "
                      "8-rank run), the grouped launches only (%s = %d bucket sets per launch); the --stats average in %s_%s_kernel_stats.csv also "
                      "covers the %d single-MSM launches of the latency measurement" % (tag, suffix, env, what, wmax, tag, suffix, len(rows) - len(grouped))}
    with open(os.path.join(prof, "rocprof_kernel_ms.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
