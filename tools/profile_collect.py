#!/usr/bin/env python3
"""Condense a tools/profile_round.sh run into the tracked profiles/ directory.
usage: python tools/profile_collect.py gpurun_out/prof_<tag> <out_prefix>      e.g. ... gpurun_out/prof_r01_v7 profiles/r01_v7
Writes <prefix>_kernel_stats.csv (rocprofv3 --stats, verbatim) and <prefix>_pmc_{fetch,write,sq}_per_kernel.csv,
<prefix>_cal_{fetch,write}_per_kernel.csv (per kernel and counter: calls, average, min, max of the raw counter values)."""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict


def condense(dirpath, out):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    if not acc:
        return False
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "calls", "avg", "min", "max"])
        for (k, c), v in sorted(acc.items()):
            w.writerow([k, c, len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % min(v), "%.1f" % max(v)])
    return True


def main():
    root, prefix = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], prefix + "_kernel_stats.csv")
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "cal_fetch", "cal_write"):
        if condense(os.path.join(root, sub), "%s_%s_per_kernel.csv" % (prefix, sub)):
            print("wrote %s_%s_per_kernel.csv" % (prefix, sub))


if __name__ == "__main__":
    main()
