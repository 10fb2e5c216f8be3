"""Batch throughput (BASELINE config 5 shape: many 2^18 MSMs over one base) by bases mode and MSMs per launch.
Usage: python tools/endo_batch_groups.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << logn
ctx = m.MsmContext(0)
ctx.set_stage_timing(0)
pts = ctx.sample_points(n, 1)
sc = torch.cat([ctx.sample_scalars(n, 2 + i) for i in range(8)], dim=0).contiguous()
total = 64
for mode in ("plain", "endomorphism", "tables", "tables_wide"):
    ctx.set_bases(pts, endomorphism=mode == "endomorphism", precompute="wide" if mode == "tables_wide" else mode == "tables")
    for g in (1, 2, 4, 8):
        if (mode == "plain" and g > 4) or (mode == "tables_wide" and g > 24 >> (ctx.wide_bits() - 16)):
            continue
        for depth in (2, 3):
            def run(count):
                fl = []
                for j in range(count // g):
                    if len(fl) == depth:
                        ctx.finish_batch(fl.pop(0), g)
                    ctx.launch_batch(sc[: g * n], n, j % 4)
                    fl.append(j % 4)
                for s in fl:
                    ctx.finish_batch(s, g)
            run(16)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(total * 2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (total * 2) * 1e3
            print("2^%d %-12s %d per launch, depth %d: %.4f ms per MSM (%.0f MSM/s)" % (logn, mode, g, depth, dt, 1e3 / dt), flush=True)
