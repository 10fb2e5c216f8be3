"""Single-MSM latency (median of 60 back-to-back synchronous calls, device-resident scalars, endomorphism bases) and its stage times, for the
environment's tuning switches.  usage: latency_probe.py [logn]; LATENCY_BASES = auto | plain | endomorphism | tables | tables_wide"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << logn
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
sc = [ctx.sample_scalars(n, 2 + i) for i in range(2)]
mode = os.environ.get("LATENCY_BASES", "auto")  # auto (the ABI's default) | plain | endomorphism | tables | tables_wide
ctx.set_bases(pts, endomorphism={"auto": None, "endomorphism": True}.get(mode, False), precompute={"tables": True, "tables_wide": "wide"}.get(mode, False))
for level in (0, 2):
    ctx.set_stage_timing(level)
    lat = []
    for i in range(80):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ctx.msm(sc[i & 1])
        lat.append((time.perf_counter() - t) * 1e3)
    lat = sorted(lat[20:])
    print(mode, "logn %d timing_level %d: median %.4f ms, best %.4f ms %s" % (logn, level, lat[len(lat) // 2], lat[0],
          {k: round(v, 3) for k, v in ctx.stage_ms().items()} if level else ""), flush=True)
ctx.close()
