#!/bin/bash
# same-box A/B: product library against the run-end-prefetch variant of k_smvp_chunks (MSM_SMVP_PREFETCH_RUNEND=1)   bash tools/r05_pf_ab.sh <variant.so>
var=$1
export BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('%-8s value %8.2f steady %8.2f smvp_ms %.4f steady_smvp %.4f lat %.3f' % ('$tagv', d['value'], d['value_steady_state'], d['roofline']['kernel_ms'], d['steady_state']['smvp_kernel_ms'], d['latency_ms_single_msm'] or 0))"; }
for k in 1 2 3; do
  for tagv in product variant; do
    ( [ $tagv = variant ] && export MSM_HIP_SO=$var; echo -n "2^20 whole   "; one --steps 20 --warmup 5 )
  done
done
for k in 1 2; do
  for tagv in product variant; do
    ( [ $tagv = variant ] && export MSM_HIP_SO=$var; export BENCH_EMULATE_WORLD=8; echo -n "share of 8   "; one --steps 64 --warmup 8 )
  done
done
for k in 1 2; do
  for tagv in product variant; do
    ( [ $tagv = variant ] && export MSM_HIP_SO=$var; echo -n "2^16 whole   "; one --steps 48 --warmup 6 --logn 16 )
  done
done
