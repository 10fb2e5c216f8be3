#!/usr/bin/env python3
"""A/B of the fixed-base tables (MSM_HIP_BASES_PRECOMPUTE) against the plain engine on one box: pipelined throughput and
single-MSM latency at 2^20, and BASELINE config 5's shape (batch of 2^18-point MSMs over one base).
usage: python tools/precompute_ab.py > gpurun_out/r02_precompute_ab.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402

ctx = m.MsmContext(0)


def pipelined(sets, steps=60):
    ctx.set_stage_timing(1)
    ctx.launch(sets[0], 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    smvp = 0.0
    for i in range(1, steps):
        ctx.launch(sets[i & 1], i & 1)
        ctx.finish((i - 1) & 1)
        smvp += ctx.stage_ms()["smvp"]
    ctx.finish((steps - 1) & 1)
    return (time.perf_counter() - t0) * 1e3 / steps, smvp / (steps - 1)


def latency(s):
    ctx.set_stage_timing(2)
    lat = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.msm(s)
        lat.append((time.perf_counter() - t0) * 1e3)
    return sorted(lat)[3], ctx.stage_ms()


print("fixed-base tables (16 x base memory, one bucket set for all 16 windows) vs plain engine, MI355X, same box")
for logn in (20, 18, 16):
    n = 1 << logn
    pts = ctx.sample_points(n, 1)
    sets = [ctx.sample_scalars(n, 2 + k) for k in range(2)]
    for mode in ("plain", "tables"):
        t0 = time.perf_counter()
        ctx.set_bases(pts, precompute=(mode == "tables"))
        setup = (time.perf_counter() - t0) * 1e3
        for _ in range(2):
            pipelined(sets, 10)
        step, smvp = pipelined(sets)
        lat, st = latency(sets[0])
        print("2^%d %-6s set_bases %7.1f ms | pipelined %.3f ms/MSM (SMVP kernel %.3f) | latency %.3f ms (smvp %.3f stitch %.3f reduce %.3f)"
              % (logn, mode, setup, step, smvp, lat, st["smvp"], st["smvp_stitch"], st["bucket_reduce"]), flush=True)
# config 5 shape on one GPU: 16 MSMs of 2^18 over one base through the library's batch runner
n, batch = 1 << 18, 16
pts = ctx.sample_points(n, 5)
sc = ctx.sample_scalars(n * batch, 6)
ctx.set_stage_timing(0)
for mode in ("plain", "tables"):
    ctx.set_bases(pts, precompute=(mode == "tables"))
    ctx.msm_batch(sc, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.msm_batch(sc, n)
    per = (time.perf_counter() - t0) * 1e3 / (3 * batch)
    print("batch 16 x 2^18 %-6s %.3f ms per MSM (%d MSMs per launch)" % (mode, per, ctx.batch_group_size(n)), flush=True)
