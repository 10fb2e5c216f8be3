#!/bin/bash
# Round-5 sort experiments (run on the GPU box from the repo root): per-kernel rocprofv3 averages of short bench runs.
#   bash tools/r05_prof_sort.sh <tag> <name>:<env assignments...>:<bench args> ...
# e.g. bash tools/r05_prof_sort.sh a "base24::--logn 24" "planes24:MSM_HIP_PLANES_WHOLE=1:--logn 24"
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$tag
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
export BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=${rest#*:}
  (
    for e in $envs; do export "$e"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline $args > $OUT/$name.log 2>&1
  ) || { echo "$name failed"; tail -5 $OUT/$name.log; exit 2; }
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${name}_kernel_stats.csv
  echo "== $name ($envs | $args)"
  grep '^{' $OUT/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('   value %.1f ms/step %.4f steady %.4f smvp %.4f' % (d['value'], d['ms_per_step'], d['ms_per_step_steady_state'], d['roofline']['kernel_ms']))"
  python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print('   %-60s calls %5s avg_us %10.1f total_ms %9.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
