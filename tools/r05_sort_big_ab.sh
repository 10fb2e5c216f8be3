#!/bin/bash
# same-box A/B of the round-5 sort for large MSMs: big-batch planes scatter on / off (MSM_HIP_BIG_PLANES_LOG=40) x fine chunk 8192 / 4096 (variant library)
#   bash tools/r05_sort_big_ab.sh <fc4k.so> <logn> [rounds]
fc=$1; logn=$2; rounds=${3:-2}
export BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0
for k in $(seq $rounds); do
  for v in big_fc8k old_fc8k big_fc4k old_fc4k; do
    ( case $v in *fc4k) export MSM_HIP_SO=$fc;; esac; case $v in old*) export MSM_HIP_BIG_PLANES_LOG=40;; esac
      python bench.py --steps 12 --warmup 4 --no-cpu-baseline --logn $logn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d['stage_ms_single_msm']
print('2^$logn %-9s value %8.2f steady %8.2f lat %.3f | count %.3f scan %.3f scatter %.3f fine %.3f (sort %.3f) smvp %.3f' % ('$v', d['value'], d['value_steady_state'], d['latency_ms_single_msm'], s['recode_count'], s['coarse_scan'], s['coarse_scatter'], s['fine_sort'], s['recode_count'] + s['coarse_scan'] + s['coarse_scatter'] + s['fine_sort'], s['smvp']))" )
  done
done
