#!/usr/bin/env python3
"""Scope C (one-shot: bases + scalars from pageable host memory) under the overlap switches of round 5, and the raw H2D rates behind it.
usage: python tools/r05_oneshot_probe.py [logn]      (environment: MSM_HIP_ONESHOT_OVERLAP, MSM_HIP_ONESHOT_CHUNK_LOG are read by the library)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
L = m.lib()
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
sc = ctx.sample_scalars(n, 2)
pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
ctx.set_bases(pts, endomorphism=True)
want = ctx.msm(sc)
out = C.create_string_buffer(96)
ts = []
for k in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.msm_hip_msm_bn254_g1(pb, sb, n, out)
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0 and m.G1(out.raw) == want
ts = sorted(ts[2:])
# raw H2D from pageable memory: one call against chunks
dst = torch.empty(len(pb), dtype=torch.uint8, device="cuda")
src = torch.frombuffer(bytearray(pb), dtype=torch.uint8)
def h2d(chunk):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o in range(0, len(pb), chunk):
        dst[o:o + chunk].copy_(src[o:o + chunk], non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
h2d(len(pb))
rates = {c >> 20: min(h2d(c) for _ in range(3)) for c in (len(pb), len(pb) // 2, len(pb) // 8, len(pb) // 32)}
print("2^%d overlap=%s chunk_log=%s: one-shot median %.3f min %.3f ms | pageable H2D of %d MiB in chunks of MiB -> ms: %s" % (
    logn, os.environ.get("MSM_HIP_ONESHOT_OVERLAP", "1"), os.environ.get("MSM_HIP_ONESHOT_CHUNK_LOG", "17"), ts[len(ts) // 2], ts[0], len(pb) >> 20,
    {k: round(v, 3) for k, v in rates.items()}), flush=True)
