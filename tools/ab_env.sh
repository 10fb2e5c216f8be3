#!/bin/bash
# Same-box A/B of an environment switch of the library: bash tools/ab_env.sh VAR=value [bench args]   ("new" = without it, "other" = with it)
setting=$1; shift
args=${@:---steps 20 --warmup 5}
for rnd in 1 2 3; do
  for name in new other; do
    if [ $name = other ]; then pre="env $setting"; else pre=""; fi
    $pre python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$rnd $name value %.1f ms %.4f smvp_ms %.4f frac %.4f lat %.3f stages %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['latency_ms_single_msm'] or 0, {k: round(v, 3) for k, v in (d.get('stage_ms_single_msm') or {}).items()}))"
  done
done
