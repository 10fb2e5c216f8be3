"""Would the endomorphism pay for a window-sharded rank?  Its grouped launch (g MSMs x its half-windows over 2n points) has the
device work of g vectors x fewer windows over 2n points through the plain entry point (scalar reads over-estimated: 32 B
instead of 16 B per input; the split pre-pass is not included: + ~0.1 ms per 8 x 2^20 scalars).
usage: endo_share_estimate.py [logn]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
ctx = m.MsmContext(0)
ctx.set_stage_timing(0)
pts = ctx.sample_points(2 * n, 1)
sc1 = torch.cat([ctx.sample_scalars(n, 2 + i) for i in range(8)], dim=0).contiguous()
sc2 = torch.cat([ctx.sample_scalars(2 * n, 12 + i) for i in range(8)], dim=0).contiguous()


def pipelined(scal, npts, g, wb, we, reps=40, depth=3):
    outs = [torch.empty((g * (we - wb), 96), dtype=torch.uint8, device="cuda") for _ in range(4)]
    def run(k):
        fl = []
        for j in range(k):
            if len(fl) == depth:
                ctx.slot_sync(fl.pop(0))
            ctx.launch_windows_batch(scal[: g * npts], npts, wb, we, j % 4, outs[j % 4], inputs_complete=True)
            fl.append(j % 4)
        for s in fl:
            ctx.slot_sync(s)
    run(6); torch.cuda.synchronize(); t0 = time.perf_counter(); run(reps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for world in (8, 4, 2):
    per = 16 // world
    g = world
    ctx.set_bases(pts[:n].contiguous())
    a = pipelined(sc1, n, g, 0, per)
    ctx.set_bases(pts)
    b = pipelined(sc2, 2 * n, g, 0, per // 2)
    print("share of %d ranks, %d MSMs per launch: plain %d windows x n: %.4f ms   %d windows x 2n: %.4f ms  (%.1f %%)  per MSM %.4f -> %.4f" %
          (world, g, per, a, per // 2, b, 100 * (b / a - 1), a / g, b / g), flush=True)
