#!/bin/bash
# one rank's share of an 8-rank run at 2^20, emulated on this GPU: plain 16-bit windows against the virtual windows of wide tables
out=${1:-gpurun_out/r05_share_ab.txt}; steps=${2:-64}
show() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-10s value %.1f ms %.4f steady %.4f smvp %.4f  w/gpu %s g %s' % ('$1', d['value'], d['ms_per_step'], d['ms_per_step_steady_state'], d['roofline']['kernel_ms'], d['config']['windows_per_gpu'], d['config']['msms_per_launch']))"; }
export BENCH_CONFIGS=0 BENCH_TABLES_WIDE=0 BENCH_EMULATE_WORLD=8
for k in 1 2; do
  python bench.py --steps $steps --warmup 8 --no-cpu-baseline 2>/dev/null | show plain >> $out
  for b in ${BITS:-19 20}; do BENCH_BASES=tables_wide BENCH_WIDE_BITS=$b python bench.py --steps $steps --warmup 8 --no-cpu-baseline 2>/dev/null | show wide$b >> $out; done
done
cat $out
