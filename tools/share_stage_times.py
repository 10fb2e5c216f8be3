"""Isolated stage times of ONE grouped launch of a window-sharded rank (nvec MSMs x its windows), against a whole MSM.
usage: share_stage_times.py [logn]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
sc = torch.cat([ctx.sample_scalars(n, 2 + i) for i in range(16)], dim=0).contiguous()
ctx.set_bases(pts, endomorphism=True)
ctx.set_stage_timing(2)
# (vectors, window range, half-length windows of the endomorphism split?)
for nvec, wb, we, halves in ((1, 0, 16, False), (8, 0, 2, False), (8, 14, 16, False), (16, 0, 2, False), (4, 0, 4, False), (2, 0, 8, False), (1, 0, 2, False),
                             (1, 0, 8, True), (8, 0, 1, True), (16, 0, 1, True), (4, 0, 2, True)):
    out = torch.empty((nvec * (we - wb), 96), dtype=torch.uint8, device="cuda")
    best = None
    launch = ctx.launch_half_windows_batch if halves else ctx.launch_windows_batch
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        launch(sc[: nvec * n], n, wb, we, 0, out, inputs_complete=True)
        ctx.slot_sync(0)
        dt = (time.perf_counter() - t0) * 1e3
        st = ctx.stage_ms()
        if best is None or dt < best[0]:
            best = (dt, st)
    print("%2d MSMs x %s windows [%d, %d): %6.3f ms  %s" % (nvec, "half" if halves else "full", wb, we, best[0], {k: round(v, 3) for k, v in best[1].items() if k != "host_finalise"}), flush=True)
