#!/bin/bash
# Round 4: kernel stats of the pipelined bench in the wide fixed-base mode at several digit widths (run on the GPU box from the repo root):
#   bash tools/r04_wide_prof.sh [logn] [steps] [widths]
logn=${1:-22}; steps=${2:-12}; widths=${3:-"17 19 20"}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BENCH_BASES=tables_wide
for b in $widths; do
  export MSM_HIP_WIDE_BITS=$b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_wide_w${b}_logn$logn -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 5 --logn $logn --no-cpu-baseline > $out/wide_trace_w${b}_logn$logn.log 2>&1
  f=$(find $out/prof_wide_w${b}_logn$logn -name "*kernel_stats.csv" | head -1)
  echo "== 2^$logn, $b-bit digits"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:13]:
    print("%-60s calls %6s avg_us %10.1f total_ms %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
