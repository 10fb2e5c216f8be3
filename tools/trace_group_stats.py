#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 kernel trace restricted to the launches of ONE shape: the kernels between two consecutive k_smvp_chunks
launches whose grid has the given number of local windows (grid y) are attributed to that launch.  usage: trace_group_stats.py <kernel_trace.csv> <grid_y>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msmk::", ""), int(r["Grid_Size_Y"]), int(r["Grid_Size_X"])))
rows.sort()
want = int(sys.argv[2])
# a kernel belongs to a launch of the wanted shape if its own grid y is the launch's local-window count (sort_fine, smvp, stitch, rowcol) or,
# for the others, if the nearest smvp launch in time has it
smvp = [(s, e) for s, e, n, gy, gx in rows if n == "k_smvp_chunks" and gy == want]
if not smvp:
    raise SystemExit("no k_smvp_chunks launch with grid y = %d" % want)
acc = defaultdict(list)
for s, e, n, gy, gx in rows:
    if n in ("k_sample_points", "k_sample_scalars", "k_precompute_tables", "k_convert_points", "k_endo_points"):
        continue
    near = min(smvp, key=lambda t: abs(t[0] - s))
    other = [t for t in [(s2, e2) for s2, e2, n2, gy2, gx2 in rows if n2 == "k_smvp_chunks"] if abs(t[0] - s) < abs(near[0] - s)]
    if other:
        continue
    acc[n].append((e - s) / 1e3)
tot = 0.0
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    print("   %-28s calls %4d avg_us %9.1f" % (n, len(v), sum(v) / len(v)))
    tot += sum(v) / len(v)
print("   sum of averages %.1f us over %d launches" % (tot, len(smvp)))
