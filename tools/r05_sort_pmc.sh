#!/bin/bash
# WRITE_SIZE / FETCH_SIZE per kernel (separate passes) of the sort kernels, big-batch planes scatter on and off    bash tools/r05_sort_pmc.sh <logn>
logn=${1:-22}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_sortpmc
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0
for v in big old; do
  for c in WRITE_SIZE FETCH_SIZE; do
    ( [ $v = old ] && export MSM_HIP_BIG_PLANES_LOG=40
      timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/${v}_$c -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --logn $logn > $OUT/${v}_$c.log 2>&1 ) || echo "$v $c failed"
    f=$(find $OUT/${v}_$c -name "*counter_collection.csv" | head -1)
    python3 - $f $v $c <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('msmk::','')
    if any(x in k for x in ('k_count','k_scatter','k_sort_fine','k_fine_hist')):
        acc[k].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print('%-4s %-10s %-34s launches %4d avg %12.1f KiB' % (sys.argv[2], sys.argv[3], k, len(v), sum(v)/len(v)))
PY
  done
done
