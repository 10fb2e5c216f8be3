"""Host-side cost per step of ShardedMsmPipeline (1 rank, no process group): tiny n so the device is never the limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
from msm_webgpu_amd.sharding import ShardedMsmPipeline, window_range
ctx = m.MsmContext(0)
n = 1024
pts, sc = ctx.sample_points(n, 1), ctx.sample_scalars(n, 2)
ctx.set_bases(pts)
pipe = ShardedMsmPipeline(ctx, 0, 1)
pipe.w_begin, pipe.w_end = window_range(0, 8)
K = 300
import cProfile, pstats
def loop():
    inflight = 0
    for i in range(K):
        pipe.issue(sc); inflight += 1
        if inflight == pipe.depth:
            pipe.complete(); inflight -= 1
    while inflight:
        pipe.complete(); inflight -= 1
loop()
torch.cuda.synchronize()
t0 = time.perf_counter(); loop(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("sharded pipeline, n=1024, 2 windows: %.1f us per step" % ((t1 - t0) / K * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
