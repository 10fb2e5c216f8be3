#!/usr/bin/env python3
"""Stage times of the two-slot pipeline UNDER OVERLAP (every stage boundary event recorded; the events cost ~8 us each, so the
step is a little longer than bench.py's) next to the same stages of isolated MSMs.  usage: python tools/pipeline_stage_times.py [logn]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
sc = [ctx.sample_scalars(n, 2 + k) for k in range(2)]
ctx.set_bases(pts)
names = ["recode_count", "coarse_scan", "coarse_scatter", "fine_sort", "smvp", "smvp_stitch", "bucket_reduce", "device_total"]


def run(level, steps=60):
    ctx.set_stage_timing(level)
    acc = {k: 0.0 for k in names}
    ctx.launch(sc[0], 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, steps):
        ctx.launch(sc[i & 1], i & 1)
        ctx.finish((i - 1) & 1)
        st = ctx.stage_ms()
        for k in names:
            acc[k] += st[k]
    ctx.finish((steps - 1) & 1)
    step = (time.perf_counter() - t0) * 1e3 / steps
    return step, {k: v / (steps - 1) for k, v in acc.items()}


for _ in range(2):
    run(0, 10)
step0, _ = run(0)
step1, st1 = run(1)
step2, st2 = run(2)
iso = {k: 0.0 for k in names}
ctx.set_stage_timing(2)
for i in range(8):
    ctx.msm(sc[i & 1])
    st = ctx.stage_ms()
    for k in names:
        iso[k] += st[k] / 8
print("2^%d: pipelined step %.3f ms (no stage events), %.3f (SMVP events only), %.3f (all stage events)" % (logn, step0, step1, step2))
print("%-16s %10s %10s" % ("stage", "pipelined", "isolated"))
for k in names:
    print("%-16s %10.3f %10.3f" % (k, st2[k], iso[k]))
print("main-stream sum pipelined %.3f  isolated %.3f" % (sum(st2[k] for k in names[:5]), sum(iso[k] for k in names[:5])))
