"""Randomised differential test of the HIP engine against the CPU oracle: random sizes (tile / chunk / alignment edges),
random scalar distributions (uniform, few distinct values, small values, zeros, equal), random window ranges.
Usage: python tools/fuzz_gpu.py [cases] [seed] [bn254|grumpkin|pallas|vesta]   (test infrastructure: uses the oracle)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
from msm_webgpu_amd.sharding import window_range
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
curve = sys.argv[3] if len(sys.argv) > 3 else "bn254"
if curve != "bn254":  # another curve through the same entry points (its oracles: oracle/cpu_<curve>.py, <curve>_ref.py)
    import importlib
    cpu, ref = importlib.import_module("oracle.cpu_" + curve), importlib.import_module("oracle." + curve + "_ref")
else:
    from oracle import cpu, bn254_ref as ref
rnd = random.Random(seed)
ctx = m.MsmContext(0, curve=curve)
combine = lambda sums: m.MsmContext.combine_windows(sums, curve=curve)
mg = {}  # lazily created msm_hip_mgpu handles by rank count (several contexts on this one GPU, pinned-buffer gather)
R = ref.R
special_n = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8193, 16385, 32769, 65535, 65537]
t0 = time.time()
for case in range(cases):
    n = rnd.choice(special_n) if rnd.random() < 0.4 else rnd.randrange(1, 70000)
    pseed, sseed = rnd.randrange(1 << 30), rnd.randrange(1 << 30)
    points = cpu.sample_points(pseed, n)
    base = ref.bytes_to_scalars(cpu.sample_scalars(sseed, n))
    kind = rnd.choice(["uniform", "few", "small", "zeros", "equal", "dup_points", "edge"])
    if kind == "few":
        vals = [base[rnd.randrange(n)] for _ in range(rnd.randrange(1, 6))]
        sc = [rnd.choice(vals) for _ in range(n)]
    elif kind == "small":
        bits = rnd.choice([1, 2, 8, 15, 16, 17, 31, 33])
        sc = [rnd.randrange(1 << bits) for _ in range(n)]
    elif kind == "zeros":
        sc = [0 if rnd.random() < 0.7 else base[i] for i in range(n)]
    elif kind == "equal":
        sc = [base[0]] * n
    elif kind == "edge":
        pool = [0, 1, R - 1, R - 2, 0x8000, 0x7FFF, 0xFFFF, int("8000" * 15, 16), (1 << 250) - 1, int("7fff" * 15, 16), (1 << 253) + 12345]
        sc = [rnd.choice(pool) for _ in range(n)]
    else:
        sc = base
    if kind == "dup_points":
        pts = ref.bytes_to_points(points)
        k = max(1, n // 3)
        pts = [pts[i % k] if rnd.random() < 0.8 else ref.neg(pts[i % k]) for i in range(n)]
        points = ref.points_to_bytes(pts)
        sc = [base[i % max(1, n // 5)] for i in range(n)]
    sb = ref.scalars_to_bytes(sc)
    want = cpu.to_affine64(cpu.cpu_msm(points, sb))
    mode = rnd.choice(["host", "device", "windows", "batch", "group", "hostbatch", "mont", "bits", "tables", "tables_batch", "hostpipe", "mgpu", "endo", "endo_batch"])
    ctx.set_bases(points, precompute=mode.startswith("tables"), endomorphism=mode.startswith("endo"))
    if mode == "mont":
        # both inputs as R = 2^256 Montgomery words (MSM_HIP_BASES_MONT256, MSM_HIP_SCALARS_MONT256)
        PM, RM = ref.P, ref.R
        pm = b"".join(((x << 256) % PM).to_bytes(32, "little") + ((y << 256) % PM).to_bytes(32, "little") for x, y in ref.bytes_to_points(points))
        sm = b"".join(((v << 256) % RM).to_bytes(32, "little") for v in ref.bytes_to_scalars(sb))
        ctx.set_bases(pm, mont256=True)
        ctx.set_scalar_format(True)
        try:
            got = ctx.msm(sm)
        finally:
            ctx.set_scalar_format(False)
    elif mode == "host" or mode == "tables":
        got = ctx.msm(sb)
    elif mode == "bits":
        # every window size, host and device scalars
        bits = rnd.choice([12, 14, 16])
        ctx.set_window_bits(bits)
        try:
            got = ctx.msm(sb) if rnd.random() < 0.5 else ctx.msm(torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda())
            assert ctx.last_window_bits() == bits
        finally:
            ctx.set_window_bits(0)
    elif mode == "endo":
        # endomorphism halves, every window size, host and device scalars
        bits = rnd.choice([0, 12, 14, 16])
        ctx.set_window_bits(bits)
        try:
            got = ctx.msm(sb) if rnd.random() < 0.5 else ctx.msm(torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda())
        finally:
            ctx.set_window_bits(0)
    elif mode == "endo_batch":
        k = rnd.randrange(1, 11)
        got = ctx.msm_batch(sb * k, n)[k - 1]
    elif mode == "tables_batch":
        k = rnd.randrange(1, 7)
        got = ctx.msm_batch(sb * k, n)[k - 1]
    elif mode == "hostpipe":
        # host scalars through the copy stream, three slots in flight
        for slot in range(3):
            ctx.launch_host(sb, slot)
        outs = [ctx.finish(slot) for slot in range(3)]
        assert outs[0] == outs[1] == outs[2]
        got = outs[2]
    elif mode == "mgpu":
        world = rnd.choice([1, 2, 3, 5, 8])
        if world not in mg:
            mg[world] = m.MultiGpuMsm([0] * world, "host", curve=curve)
        mg[world].set_bases(points)
        got = mg[world].msm(sb) if rnd.random() < 0.6 else mg[world].msm_batch(sb + sb, n)[1]
    elif mode == "device":
        t = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
        got = ctx.msm(t)
    elif mode == "batch":
        t = torch.frombuffer(bytearray(sb + sb), dtype=torch.uint8).cuda()
        got = ctx.msm_batch(t, n)[1]
    elif mode == "hostbatch":
        got = ctx.msm_batch(sb + sb + sb, n)[2]
    elif mode == "group":
        # several MSMs per launch (the multi-GPU pipeline's grouped launches): vector `pos` of the group is this case's,
        # the others are all-zero / copies; every rank's share is run and the window sums are combined on the host
        world = rnd.choice([2, 4, 8, 16])
        g = 16 // (16 // world)
        pos = rnd.randrange(g)
        vecs = [bytes(len(sb)) if rnd.random() < 0.5 else sb for _ in range(g)]
        vecs[pos] = sb
        t = torch.frombuffer(bytearray(b"".join(vecs)), dtype=torch.uint8).cuda()
        parts = []
        for r in range(world):
            b, e = window_range(r, world)
            out = torch.empty((g * (e - b), 96), dtype=torch.uint8, device="cuda")
            ctx.launch_windows_batch(t, n, b, e, r % 4, out)
            ctx.slot_sync(r % 4)
            parts.append(out[pos * (e - b):(pos + 1) * (e - b)])
        got = combine(torch.cat(parts, dim=0))
    else:
        t = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
        world = rnd.choice([2, 3, 4, 5, 8, 16])
        parts = [ctx.msm_windows(t, *window_range(r, world)) for r in range(world)]
        got = combine(torch.cat(parts, dim=0))
    if (case + 1) % 25 == 0:
        print("  %d cases ok, %.0f s" % (case + 1, time.time() - t0), flush=True)
    if got.to_affine_bytes() != want:
        print("MISMATCH case", case, "n", n, "kind", kind, "mode", mode, "seeds", pseed, sseed)
        sys.exit(1)
print("fuzz ok: %d cases in %.1f s (seed %d, %s)" % (cases, time.time() - t0, seed, curve))
