#!/bin/bash
# same-box A/B of the default (endomorphism) mode's sort: per-kernel rocprofv3 averages + step times, new library against a base library
#   bash tools/r05_sort_ab.sh <tag> <logn> [base.so]
tag=$1; logn=$2; base=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$tag
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0
run() {  # name, env...
  name=$1; shift
  ( for e in "$@"; do export "$e"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --logn $logn > $OUT/$name.log 2>&1 ) || { echo "$name failed"; tail -3 $OUT/$name.log; return; }
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name logn $logn ($*)"
  python3 - $f <<'PY'
import csv,sys
rows={r['Name'].split('(')[0].replace('void ','').replace('msmk::',''):r for r in csv.DictReader(open(sys.argv[1]))}
tot=0
for k in sorted(rows):
    if any(x in k for x in ('k_count','k_scan','k_scatter','k_fine_hist','k_sort_fine','k_smvp_chunks')):
        us=float(rows[k]['AverageNs'])/1e3
        if 'smvp' not in k: tot+=us
        print('   %-34s calls %4s avg_us %9.1f' % (k, rows[k]['Calls'], us))
print('   sort kernels sum %.1f us' % tot)
PY
}
run new
[ -n "$base" ] && run base MSM_HIP_SO=$base
run new_halves MSM_HIP_SCATTER_SPLIT=0
cd $ROOT
for k in 1 2; do
  for v in new base; do
    [ $v = base ] && [ -z "$base" ] && continue
    ( [ $v = base ] && export MSM_HIP_SO=$base; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --logn $logn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$v value %.2f steady %.2f ms %.4f steady_ms %.4f smvp %.4f lat %.3f' % (d['value'], d['value_steady_state'], d['ms_per_step'], d['ms_per_step_steady_state'], d['roofline']['kernel_ms'], d['latency_ms_single_msm']))" )
  done
done
