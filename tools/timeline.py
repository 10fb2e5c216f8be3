"""Timeline of the kernels of steady-state MSMs from a rocprofv3 --kernel-trace CSV.
usage: timeline.py <kernel_trace.csv> [first_msm] [count]   (an MSM starts at each k_count)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8
count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msmk::", ""), r.get("Queue_Id", "?")) for r in rows))
starts = [i for i, k in enumerate(ks) if k[2].startswith("k_count")]
i0, i1 = starts[first], starts[first + count]
t0 = ks[i0][0]
# kernels of earlier MSMs still running are included when they end after t0
for s, e, name, q in ks:
    if e >= t0 and s <= ks[i1][0]:
        print("%9.1f %9.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
print("period: %.1f us per MSM" % ((ks[i1][0] - t0) / 1e3 / count))
