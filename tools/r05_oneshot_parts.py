#!/usr/bin/env python3
"""Scope B (msm_hip_run with host scalars) and scope C (the one-shot call: bases + scalars from pageable host memory) by the number of parts the call is split into (msm_hip_test_oneshot_parts:
sub-MSMs over ranges of the points, uploads one behind the other -- only the last part's accumulation follows the upload).
usage: python tools/r05_oneshot_parts.py [logn ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402

L = m.lib()
out = C.create_string_buffer(96)
for logn in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << logn
    ctx = m.MsmContext(0)
    pts, sc = ctx.sample_points(n, 1), ctx.sample_scalars(n, 2)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    ctx.set_bases(pts, endomorphism=True)
    want = ctx.msm(sc)
    # scope B: resident bases, host scalars, one synchronous call (msm_hip_run) -- parts = result slots of the one context
    for parts in (1, 2, 3, 4, 2, 1):
        assert L.msm_hip_test_oneshot_parts(parts, 1) == 0
        ts = []
        for k in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = ctx.msm(sb)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert got == want
        ts = sorted(ts[2:])
        print("2^%d  parts %d: host-scalar run median %.3f  min %.3f  max %.3f ms" % (logn, parts, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)
    ctx.close()
    del pts, sc
    for parts in (1, 2, 3, 4, 2, 1):
        assert L.msm_hip_test_oneshot_parts(parts, 1) == 0
        ts = []
        for k in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = L.msm_hip_msm_bn254_g1(pb, sb, n, out)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0 and m.G1(out.raw) == want
        ts = sorted(ts[2:])
        print("2^%d  parts %d: one-shot median %.3f  min %.3f  max %.3f ms" % (logn, parts, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)
        L.msm_hip_oneshot_release()
    L.msm_hip_test_oneshot_parts(0, 0)
