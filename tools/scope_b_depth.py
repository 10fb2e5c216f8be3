"""Scope B (host scalars, resident bases) as a pipeline: slots in flight x pageable / pinned caller memory, against the
device-resident rate.  usage: scope_b_depth.py [logn]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
ctx = m.MsmContext(0)
ctx.set_stage_timing(0)
pts = ctx.sample_points(n, 1)
sets = [ctx.sample_scalars(n, 2 + i) for i in range(2)]
L, h = m.lib(), ctx._h
for endo in (False, True):
    ctx.set_bases(pts, endomorphism=endo)
    pinned = [s.cpu().pin_memory() for s in sets]
    pageable = [s.cpu().numpy().tobytes() for s in sets]
    out = C.create_string_buffer(96)
    def pipe(launch, depth, k=60):
        fl = []
        for j in range(k):
            if len(fl) == depth:
                assert L.msm_hip_finish_bn254(h, fl.pop(0), out) == 0
            launch(j & 1, j % 4)
            fl.append(j % 4)
        for s in fl:
            assert L.msm_hip_finish_bn254(h, s, out) == 0
    modes = {"resident": lambda i, slot: ctx.launch(sets[i], slot),
             "pageable": lambda i, slot: L.msm_hip_launch_bn254(h, pageable[i], n, slot),
             "pinned": lambda i, slot: L.msm_hip_launch_bn254(h, C.c_char_p(pinned[i].data_ptr()), n, slot)}
    for name, fn in modes.items():
        for depth in (1, 2, 3):
            pipe(fn, depth, 8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe(fn, depth)
            torch.cuda.synchronize()
            print("endo=%d %-9s depth %d: %.4f ms per MSM" % (endo, name, depth, (time.perf_counter() - t0) / 60 * 1e3), flush=True)
