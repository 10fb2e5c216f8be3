"""Split-consistency check at very large sizes: MSM(P, s) == MSM(P[:h], s[:h]) + MSM(P[h:], s[h:]).  usage: large_check.py [logn]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
P = m.api.P
def add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = 1 << logn
ctx = m.MsmContext(0)
t0 = time.time(); pts, sc = ctx.sample_points(n, 7), ctx.sample_scalars(n, 8); print("sampled 2^%d in %.2f s" % (logn, time.time() - t0))
ctx.set_bases(pts)
t0 = time.time(); whole = ctx.msm(sc).to_affine(); print("whole MSM %.1f ms" % ((time.time() - t0) * 1e3), {k: round(v, 2) for k, v in ctx.stage_ms().items()})
h = n // 2 + 4321
first = ctx.msm(sc[:h].contiguous()).to_affine()
ctx.set_bases(pts[h:].contiguous())
second = ctx.msm(sc[h:].contiguous()).to_affine()
assert add(first, second) == whole
print("split consistency ok at 2^%d" % logn)
