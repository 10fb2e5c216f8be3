#!/bin/bash
# SMVP lanes (chunk length) sweep on one box: bash tools/lanes_sweep.sh  -> ms per MSM, SMVP kernel ms, stitch ms for several shapes
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 ms %.4f smvp %.4f lat %s stitch %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('latency_ms_single_msm'), (d.get('stage_ms_single_msm') or {}).get('smvp_stitch')))"; }
for rep in 1 2; do
for t in 393216 589824 786432; do
  MSM_HIP_TARGET_LANES=$t python bench.py --no-cpu-baseline --steps 64 --warmup 5 2>/dev/null | show "endo20 lanes $t"
  BENCH_BASES=plain MSM_HIP_TARGET_LANES=$t python bench.py --no-cpu-baseline --steps 64 --warmup 5 2>/dev/null | show "plain20 lanes $t"
  BENCH_EMULATE_WORLD=8 MSM_HIP_TARGET_LANES=$t python bench.py --no-cpu-baseline --steps 64 --warmup 8 2>/dev/null | show "share8 lanes $t"
done; done
for t in 393216 589824 786432; do
  MSM_HIP_TARGET_LANES=$t python bench.py --no-cpu-baseline --steps 20 --warmup 3 --logn 22 2>/dev/null | show "endo22 lanes $t"
  MSM_HIP_TARGET_LANES=$t python bench.py --no-cpu-baseline --steps 64 --warmup 5 --logn 18 2>/dev/null | show "endo18 lanes $t"
  MSM_HIP_TARGET_LANES=$t python bench.py --no-cpu-baseline --steps 64 --warmup 5 --logn 16 2>/dev/null | show "endo16 lanes $t"
done
