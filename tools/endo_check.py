"""First light for the endomorphism mode (MSM_HIP_BASES_ENDOMORPHISM): end-to-end against the oracle at several sizes, window
sizes, base counts larger than the launch, batches; then timing against the plain mode.  (test infrastructure: uses the oracle)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
from oracle import cpu

ctx = m.MsmContext(0)
for n, nb in ((1, 1), (5, 9), (257, 257), (4096, 5000), (70001, 70001)):
    points = cpu.sample_points(7, nb)
    sc = cpu.sample_scalars(8, n)
    want = cpu.to_affine64(cpu.cpu_msm(points[:64 * n], sc))
    ctx.set_bases(points, endomorphism=True)
    assert ctx.uses_endomorphism()
    for bits in (0, 12, 14, 16):
        ctx.set_window_bits(bits)
        got = ctx.msm(sc)
        assert got.to_affine_bytes() == want, (n, nb, bits)
        got = ctx.msm(torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda())
        assert got.to_affine_bytes() == want, (n, nb, bits, "dev")
    ctx.set_window_bits(0)
    outs = ctx.msm_batch(sc * 5, n)
    assert all(o.to_affine_bytes() == want for o in outs), (n, "batch")
    # window-sharded entry points ignore the second half
    t = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    parts = [ctx.msm_windows(t, 0, 7), ctx.msm_windows(t, 7, 16)]
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)).to_affine_bytes() == want
    print("ok n=%d bases=%d" % (n, nb), flush=True)

ctx.set_stage_timing(0)
for logn in (16, 18, 20):
    n = 1 << logn
    pts = ctx.sample_points(n, 1)
    sc = ctx.sample_scalars(n, 2)
    res = {}
    for endo in (False, True):
        ctx.set_bases(pts, endomorphism=endo)
        def run(k, depth=3):
            fl = []
            for j in range(k):
                if len(fl) == depth:
                    ctx.finish(fl.pop(0))
                ctx.launch(sc, j % 4)
                fl.append(j % 4)
            return [ctx.finish(s) for s in fl][-1]
        r = run(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(60)
        torch.cuda.synchronize()
        res[endo] = ((time.perf_counter() - t0) / 60 * 1e3, r.to_affine_bytes())
    assert res[False][1] == res[True][1]
    print("2^%d pipelined: plain %.4f ms  endomorphism %.4f ms (%.1f %%)" % (logn, res[False][0], res[True][0], 100 * (res[True][0] / res[False][0] - 1)), flush=True)
