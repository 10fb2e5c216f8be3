#!/usr/bin/env python3
"""Single-MSM latency and pipelined throughput per window size and n (chooses the thresholds of pick_window_bits, msm_hip.hip).
usage (GPU box): python tools/window_bits_latency.py > gpurun_out/r02_window_bits_latency.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402

ctx = m.MsmContext(0)
N = 1 << 20
pts = ctx.sample_points(N, 1)
sc = [ctx.sample_scalars(N, 2 + k) for k in range(2)]
ctx.set_bases(pts)
print("single-MSM latency (ms, median of 9) and pipelined one-MSM-per-launch step (ms) by window bits; MI355X")
print("%6s | %s" % ("log2 n", " | ".join("c=%-2d lat   step  reduce" % b for b in (12, 14, 16))))
for logn in range(10, 21):
    n = 1 << logn
    s = [x[:n].contiguous() for x in sc]
    row = []
    for bits in (12, 14, 16):
        ctx.set_window_bits(bits)
        ctx.set_stage_timing(2)
        for _ in range(3):
            ctx.msm(s[0])
        lat = []
        for i in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.msm(s[i & 1])
            lat.append((time.perf_counter() - t0) * 1e3)
        st = ctx.stage_ms()
        ctx.set_stage_timing(0)
        steps = 40
        ctx.launch(s[0], 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, steps):
            ctx.launch(s[i & 1], i & 1)
            ctx.finish((i - 1) & 1)
        ctx.finish((steps - 1) & 1)
        step = (time.perf_counter() - t0) * 1e3 / steps
        row.append("%5.3f %6.3f %6.3f     " % (sorted(lat)[4], step, st["bucket_reduce"] + st["smvp_stitch"]))
    print("%6d | %s" % (logn, " | ".join(row)), flush=True)
ctx.set_window_bits(0)
