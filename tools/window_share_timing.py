"""Isolated (un-pipelined) stage times of one rank's share of a window-sharded MSM.  usage: window_share_timing.py [logn] [windows...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
counts = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16]
n = 1 << logn
ctx = m.MsmContext(0)
pts, sc = ctx.sample_points(n, 1), ctx.sample_scalars(n, 2)
ctx.set_bases(pts)
ctx.set_stage_timing(2)
for w in counts:
    best = None
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.msm_windows(sc, 0, w); dt = (time.perf_counter() - t0) * 1e3
        st = ctx.stage_ms()
        if best is None or dt < best[0]:
            best = (dt, st)
    print("%2d windows: %6.3f ms  %s" % (w, best[0], {k: round(v, 3) for k, v in best[1].items() if k != "host_finalise"}))
