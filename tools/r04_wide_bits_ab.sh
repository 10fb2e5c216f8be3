#!/bin/bash
# Round 4: the wide fixed-base tables at 17 .. 20-bit digits (MSM_HIP_WIDE_BITS) against the endomorphism mode and the 16-bit tables, one box
# (run on the GPU box from the repo root):  bash tools/r04_wide_bits_ab.sh [logn] [steps] [reps] [widths]
show() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), 'smvp', round(d['roofline']['kernel_ms'],4), 'lat', round(d['latency_ms_single_msm'],3))"; }
logn=${1:-20}; steps=${2:-20}; reps=${3:-2}; widths=${4:-"17 18 19 20"}
echo "== 2^$logn, --steps $steps --warmup 5"
for rep in $(seq $reps); do
  BENCH_BASES=endomorphism python bench.py --steps $steps --warmup 5 --logn $logn --no-cpu-baseline 2>/dev/null | show "endo"
  BENCH_BASES=tables python bench.py --steps $steps --warmup 5 --logn $logn --no-cpu-baseline 2>/dev/null | show "tables16"
  for b in $widths; do
    MSM_HIP_WIDE_BITS=$b BENCH_BASES=tables_wide python bench.py --steps $steps --warmup 5 --logn $logn --no-cpu-baseline 2>/dev/null | show "wide$b"
  done
done
