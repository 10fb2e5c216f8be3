#!/usr/bin/env python3
"""Where the time of the one-shot call shape goes (scope C, ≙ the reference's compute_msm: create, upload bases, run, destroy) and
of the host-scalar path (scope B): every phase timed through the C ABI.  usage: python tools/oneshot_breakdown.py [logn]"""
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
warm = m.MsmContext(0)  # code objects loaded, HIP initialised: not part of any scope
pts = warm.sample_points(n, 1)
sc = [warm.sample_scalars(n, 2 + k) for k in range(2)]
pb, sb = pts.cpu().numpy().tobytes(), [s.cpu().numpy().tobytes() for s in sc]
warm.set_bases(pts)
warm.msm(sc[0])


def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    return (time.perf_counter() - t0) * 1e3, r


for rep in range(3):
    h = C.c_void_p()
    out = C.create_string_buffer(96)
    t_create, _ = t(lambda: L.msm_hip_ctx_create(C.byref(h), 0))
    t_bases, _ = t(lambda: L.msm_hip_set_bases_bn254(h, pb, n, 0))
    t_run1, _ = t(lambda: L.msm_hip_run_bn254(h, sb[0], n, out))
    t_run2, _ = t(lambda: L.msm_hip_run_bn254(h, sb[1], n, out))
    t_destroy, _ = t(lambda: L.msm_hip_ctx_destroy(h))
    t_one, _ = t(lambda: L.msm_hip_msm_bn254_g1(pb, sb[0], n, out))
    print("2^%d rep %d: create %.2f  set_bases(host) %.2f  first run(host scalars) %.2f  second run %.2f  destroy %.2f  | one-shot %.2f ms"
          % (logn, rep, t_create, t_bases, t_run1, t_run2, t_destroy, t_one), flush=True)
# scope B: host scalars, bases resident -- latency of one call, and two slots alternating (the copy of MSM i+1 under MSM i)
lat = sorted(t(lambda: warm.msm(sb[i & 1]))[0] for i in range(7))[3]
steps = 30
warm.launch_host(sb[0], 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, steps):
    warm.launch_host(sb[i & 1], i & 1)
    warm.finish((i - 1) & 1)
warm.finish((steps - 1) & 1)
pipe = (time.perf_counter() - t0) * 1e3 / steps
# pinned host memory: the copy is asynchronous
pin = [torch.frombuffer(bytearray(b), dtype=torch.uint8).pin_memory() for b in sb]
L.msm_hip_launch_bn254.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
warm_h = warm._h
L.msm_hip_launch_bn254(warm_h, pin[0].data_ptr(), n, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, steps):
    L.msm_hip_launch_bn254(warm_h, pin[i & 1].data_ptr(), n, i & 1)
    warm.finish((i - 1) & 1)
warm.finish((steps - 1) & 1)
pipe_pinned = (time.perf_counter() - t0) * 1e3 / steps
lat_dev = sorted(t(lambda: warm.msm(sc[i & 1]))[0] for i in range(7))[3]
print("scope B 2^%d: latency host scalars %.3f ms (device-resident %.3f) | two slots alternating: pageable %.3f ms/MSM, pinned %.3f ms/MSM"
      % (logn, lat, lat_dev, pipe, pipe_pinned))
