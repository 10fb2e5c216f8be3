#!/bin/bash
# same-box A/B, new library against a base library: throughput and the UN-pipelined stage times of one MSM (HIP events at every stage boundary:
# no other launch's stitch / reduce beside the sort kernels)      bash tools/r05_stage_ab.sh <base.so> <logn> [rounds]
base=$1; logn=$2; rounds=${3:-2}
export BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0
for k in $(seq $rounds); do
  for v in new base; do
    ( [ $v = base ] && export MSM_HIP_SO=$base; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --logn $logn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d['stage_ms_single_msm']
print('2^$logn %-4s value %8.2f steady %8.2f lat %.3f | count %.3f scan %.3f scatter %.3f fine %.3f smvp %.3f stitch %.3f reduce %.3f' % ('$v', d['value'], d['value_steady_state'], d['latency_ms_single_msm'], s['recode_count'], s['coarse_scan'], s['coarse_scatter'], s['fine_sort'], s['smvp'], s['smvp_stitch'], s['bucket_reduce']))" )
  done
done
