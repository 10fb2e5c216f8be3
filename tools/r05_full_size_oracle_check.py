#!/usr/bin/env python3
"""Whole MSMs at 2^22 and 2^24 against the CPU oracle on ALL points (test infrastructure: uses the oracle; too slow for the test suite -- there the
large sizes are checked through slices and through the window-range recombination).  usage: python tools/r05_full_size_oracle_check.py [logn ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import msm_webgpu_amd as m  # noqa: E402
from oracle import cpu  # noqa: E402

threads = os.cpu_count() or 8
for logn in [int(a) for a in sys.argv[1:]] or [22, 24]:
    n = 1 << logn
    ctx = m.MsmContext(0)
    pts, sc = ctx.sample_points(n, 0x5EED + logn), ctx.sample_scalars(n, 0xABCD + logn)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    t0 = time.perf_counter()
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, threads))
    t_cpu = time.perf_counter() - t0
    out = []
    for mode, kw in (("endomorphism", dict(endomorphism=True)), ("plain", dict(endomorphism=False)), ("tables_wide", dict(precompute="wide"))):
        ctx.set_bases(pts, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = ctx.msm(sc)
        dt = time.perf_counter() - t0
        out.append("%s %s (%.1f ms)" % (mode, "bit-exact" if got.to_affine_bytes() == want else "MISMATCH", dt * 1e3))
    print("2^%d: oracle on %d threads %.1f s | %s" % (logn, threads, t_cpu, " | ".join(out)), flush=True)
    ctx.close()
    del pts, sc, pb, sb
