"""Run a few isolated (non-pipelined) MSMs so that a rocprofv3 kernel trace shows per-kernel latency.  usage: trace_latency.py [logn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
ctx = m.MsmContext(0)
pts, sc = ctx.sample_points(n, 1), ctx.sample_scalars(n, 2)
ctx.set_bases(pts)
for _ in range(6):
    ctx.msm(sc)
print(ctx.stage_ms())
