"""What would the curve endomorphism (k = k1 + k2 lambda, 2n points x 128-bit scalars) buy?  Its device work is
that of 8 of the 16 windows over 2n points: measured here with the window-range entry point before the mode was written.
Usage: python tools/endo_estimate.py [log_n] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = m.MsmContext(0)
ctx.set_stage_timing(0)


def pipelined(n, w_begin, w_end, depth=3):
    ctx.set_bases(ctx.sample_points(n, 1))
    sc = ctx.sample_scalars(n, 2)
    outs = [torch.empty((w_end - w_begin, 96), dtype=torch.uint8, device="cuda") for _ in range(4)]
    def run(k):
        inflight = []
        for j in range(k):
            if len(inflight) == depth:
                ctx.slot_sync(inflight.pop(0))
            ctx.launch_windows_batch(sc, n, w_begin, w_end, j % 4, outs[j % 4], inputs_complete=True)
            inflight.append(j % 4)
        for s in inflight:
            ctx.slot_sync(s)
    run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(reps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = 1 << logn
for depth in (1, 3):
    a = pipelined(n, 0, 16, depth)
    b = pipelined(2 * n, 0, 8, depth)
    print("2^%d depth %d: 16 windows x n: %.4f ms   8 windows x 2n: %.4f ms   (%.1f %%)" % (logn, depth, a, b, 100 * (b / a - 1)), flush=True)
