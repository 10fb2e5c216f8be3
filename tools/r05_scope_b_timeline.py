#!/usr/bin/env python3
"""Kernels and memory copies of the LAST host-scalar call of a tools/r05_scope_b_trace.py run, on one time axis (us from the call's first copy).
usage: python tools/r05_scope_b_timeline.py <rocprofv3 output dir>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("msmk::", "").replace("void ", ""), "q" + r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), "copy"))
ev.sort()
# the last call: from the last big host-to-device copy that is preceded by a gap of > 1 ms without kernels
big = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2] and e[1] - e[0] >= 100_000]  # (the trace has no sizes: a scalar upload takes > 0.1 ms)
starts = [i for k, i in enumerate(big) if k == 0 or ev[i][0] - ev[big[k - 1]][1] > 1_000_000]
i0 = starts[-1]
t0 = ev[i0][0]
for s, e, name, q in ev[i0:]:
    print("%9.1f %9.1f  %8.1f us  %-5s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
