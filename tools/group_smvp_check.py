"""SMVP kernel time for 16 local windows formed in different ways (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
n = 1 << 20
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
ctx.set_bases(pts)
vecs = [ctx.sample_scalars(n, 10 + k) for k in range(8)]
batch = torch.cat(vecs, dim=0).contiguous()
same = torch.cat([vecs[0]] * 8, dim=0).contiguous()
out = torch.empty((16, 96), dtype=torch.uint8, device="cuda")
ctx.set_stage_timing(2)
def run(name, sc, nn, b, e):
    best = None
    for _ in range(4):
        ctx.launch_windows_batch(sc, nn, b, e, 0, out); ctx.slot_sync(0)
        st = ctx.stage_ms()
        if best is None or st["smvp"] < best["smvp"]:
            best = st
    print("%-34s %s" % (name, {k: round(v, 3) for k, v in best.items() if k not in ("host_finalise",)}))
run("1 vector x windows 0..15", vecs[0], n, 0, 16)
run("8 distinct vectors x windows 0..1", batch, n, 0, 2)
run("8 distinct vectors x windows 6..7", batch, n, 6, 8)
run("8 distinct vectors x windows 14..15", batch, n, 14, 16)
run("8 x same vector x windows 0..1", same, n, 0, 2)
run("2 distinct vectors x windows 0..7", batch[: 2 * n], n, 0, 8)
