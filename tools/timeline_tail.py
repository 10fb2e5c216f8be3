"""The last kernels (and copies) of a rocprofv3 --kernel-trace [--memory-copy-trace] run, as a timeline: for a bench.py run that is its timed region.
usage: timeline_tail.py <dir with *_kernel_trace.csv> [launches]   (a launch starts at each k_count; default: the last 4)"""
import csv, glob, os, sys
d = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ks = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("msmk::", "").replace("void ", ""), "q" + r.get("Queue_Id", "?"), r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), "", r.get("Bytes", "")))
ks.sort()
starts = [i for i, k in enumerate(ks) if k[2].startswith("k_count")]
i0 = starts[-launches]
t0 = ks[i0][0]
for s, e, name, q, g in ks:
    if e >= t0:
        print("%9.1f %9.1f  %7.1f us  %-4s %-28s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name, g))
