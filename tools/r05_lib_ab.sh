#!/bin/bash
# same-box A/B: product library against another build (a variant, or the previous product as the base)   bash tools/r05_lib_ab.sh <other.so>
# whole MSMs at 2^20 (default and plain bases) and 2^16, one rank's share of 8 (plain and wide-table shares)
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
for k in 1 2; do
  for tagv in product variant; do
    ( [ $tagv = variant ] && export MSM_HIP_SO=$var; export BENCH_BASES=plain; echo -n "2^20 plain   "; one --steps 20 --warmup 5 )
  done
done
for k in 1 2; do
  for tagv in product variant; do
    ( [ $tagv = variant ] && export MSM_HIP_SO=$var; export BENCH_EMULATE_WORLD=8 BENCH_BASES=tables_wide; echo -n "wide share 8 "; one --steps 64 --warmup 8 )
  done
done
