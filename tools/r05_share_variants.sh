#!/bin/bash
# same-box A/B of the share-of-8 emulation (wide tables, virtual windows): library variants (MSM_HIP_SO) and environment switches
#   bash tools/r05_share_variants.sh <out> "<name>:<env assignments>" ...
out=$1; shift
show() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-22s value %.1f ms %.4f steady %.4f smvp %.4f' % ('$1', d['value'], d['ms_per_step'], d['ms_per_step_steady_state'], d['roofline']['kernel_ms']))"; }
export BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0 BENCH_EMULATE_WORLD=8
for k in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( for e in $envs; do export "$e"; done; python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | show $name >> $out )
  done
done
cat $out
