#!/bin/bash
# Round 4: kernel stats of the pipelined bench at a small size, by base mode (run on the GPU box from the repo root):  bash tools/r04_small_prof.sh [logn] [modes]
logn=${1:-16}; modes=${2:-"endomorphism tables_wide"}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BENCH_TABLES_WIDE=0
for mode in $modes; do
  export BENCH_BASES=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_small_${mode}_logn$logn -- python3 $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 6 --logn $logn --no-cpu-baseline > $out/small_trace_${mode}_logn$logn.log 2>&1
  f=$(find $out/prof_small_${mode}_logn$logn -name "*kernel_stats.csv" | head -1)
  echo "== 2^$logn, $mode: $(grep -o '"value": [0-9.]*' $out/small_trace_${mode}_logn$logn.log | head -1) MSM/s under the profiler"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-62s calls %6s avg_us %9.1f total_ms %8.2f  %4.1f%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
done
