"""Host-side cost of enqueueing one MSM (HIP API calls + ctypes), measured with a tiny n so the device is never the limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
ctx = m.MsmContext(0)
n = 1 << 20
pts, sc = ctx.sample_points(n, 1), ctx.sample_scalars(n, 2)
ctx.set_bases(pts)
ctx.msm(sc)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for i in range(K):
    ctx.launch(sc, i & 3)          # enqueue only; slots reused without finish -> measures pure enqueue cost (device will lag)
    if i >= 3:
        pass
t1 = time.perf_counter()
print("enqueue-only: %.1f us per MSM (device lags behind)" % ((t1 - t0) / K * 1e6))
for s in range(4):
    ctx.finish(s)
small = sc[:1024].contiguous()
ctx.set_bases(pts[:1024].contiguous())
t0 = time.perf_counter()
for i in range(K):
    ctx.launch(small, 0)
    ctx.finish(0)
t1 = time.perf_counter()
print("n=1024 launch+finish round trip: %.1f us" % ((t1 - t0) / K * 1e6), ctx.stage_ms())
