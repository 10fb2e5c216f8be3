#!/usr/bin/env python3
"""Scope B (msm_hip_run, host scalars, resident bases) under rocprofv3: a few calls, so that tools/r05_scope_b_timeline.py can print the kernels and
copies of the last one.   usage (GPU box): rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/r05_scope_b_trace.py [logn] [parts]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << logn
ctx = m.MsmContext(0)
pts, sc = ctx.sample_points(n, 1), ctx.sample_scalars(n, 2)
sb = sc.cpu().numpy().tobytes()
ctx.set_bases(pts, endomorphism=True)
if parts:
    m.lib().msm_hip_test_oneshot_parts(parts, 1)
ts = []
for k in range(8):
    torch.cuda.synchronize()
    time.sleep(0.002)
    t0 = time.perf_counter()
    ctx.msm(sb)
    ts.append((time.perf_counter() - t0) * 1e3)
print("2^%d parts %d: %s" % (logn, parts, " ".join("%.3f" % t for t in ts)))
