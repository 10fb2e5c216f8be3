#!/bin/bash
# Same-box A/B of two library builds: bash tools/ab_so.sh <other.so> [bench args]   (boxes differ by +-5 %: only same-box pairs count)
other=$1; shift
args=${@:---steps 64 --warmup 5}
for rnd in 1 2 3; do
  for name in new other; do
    if [ $name = other ]; then export MSM_HIP_SO=$other; else unset MSM_HIP_SO; fi
    python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$rnd $name value %.1f ms %.4f smvp_ms %.4f frac %.4f lat %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['latency_ms_single_msm'] or 0))"
  done
done
