"""Timeline of the last grouped launches (k_smvp_chunks grid y == <windows>) of a rocprofv3 --kernel-trace run.
usage: timeline_grouped.py <dir> <windows> [launches]"""
import csv, glob, os, sys
d, want = sys.argv[1], int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ks = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("msmk::", "").replace("void ", ""), "q" + r.get("Queue_Id", "?"), r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]))
ks.sort()
sm = [i for i, k in enumerate(ks) if k[2] == "k_smvp_chunks" and k[4].endswith("x%d" % want)]
# the count kernel in front of the first of the last `launches` grouped SMVPs
first = sm[-launches]
i0 = max(i for i in range(first) if ks[i][2].startswith("k_count"))
t0 = ks[i0][0]
tend = ks[sm[-1]][1] + 1500e3
for s, e, name, q, g in ks:
    if e >= t0 and s <= tend:
        print("%9.1f %9.1f  %7.1f us  %-4s %-28s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name, g))
per = (ks[sm[-1]][0] - ks[sm[-launches]][0]) / 1e3 / (launches - 1)
print("period between grouped SMVP starts: %.1f us" % per)
