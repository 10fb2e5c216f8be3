#!/bin/bash
# C4 diagnostic: is the SMVP at 2^24 / 2^22 latency-bound?  The same launch with 3 and with 2 waves per SIMD (MSM_HIP_SMVP_LDS_PAD).
for l in 20 22 24; do
  for pad in 0 65536; do
    MSM_HIP_SMVP_LDS_PAD=$pad python bench.py --logn $l --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('logn $l pad $pad value %.2f ms %.4f smvp_ms %.4f frac %.4f lat %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['latency_ms_single_msm'] or 0))"
  done
done
