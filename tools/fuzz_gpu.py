"""Randomised differential test of the HIP engine against the CPU oracle: random sizes (tile / chunk / alignment edges),
random scalar distributions (uniform, few distinct values, small values, zeros, equal), random window ranges.
Usage: python tools/fuzz_gpu.py [cases] [seed] [bn254|grumpkin|pallas|vesta|bls12_381|bn254_g2|bls12_381_g2]   (test infrastructure: uses the oracle)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
from msm_webgpu_amd.sharding import window_range
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
curve = sys.argv[3] if len(sys.argv) > 3 else "bn254"
if curve.endswith("_g2"):
    # G2 (coordinates in Fq2): the checker is the Python model's CLOSED FORM -- the points are drawn from a pool of known multiples m G of the
    # generator (and their negatives), so sum_i s_i P_i = (sum_i s_i m_i mod r) G for any size and scalar distribution
    import importlib
    ref = importlib.import_module("oracle." + curve + "_ref")

    class ClosedFormG2:
        POOL = 1 << 12

        def __init__(self):
            self.c_model = importlib.import_module("oracle.cpu_" + curve)
            self.pts, self.mult = ref.sample_points(self.POOL, 99), ref.sample_multipliers(self.POOL, 99)
            self.enc = [ref.points_to_bytes([p]) for p in self.pts]
            self.m_of = {e: k for e, k in zip(self.enc, self.mult)}
            self.m_of.update({ref.points_to_bytes([ref.neg(p)]): (-k) % ref.R for p, k in zip(self.pts, self.mult)})

        def sample_points(self, seed, n):
            r = random.Random(seed)
            start, stride = r.randrange(self.POOL), r.choice([1, 3, 5, 7])
            return b"".join(self.enc[(start + stride * i) % self.POOL] for i in range(n))

        def sample_scalars(self, seed, n):
            return ref.scalars_to_bytes([ref.sample_scalar(seed, i) for i in range(n)])

        def cpu_msm(self, points, sb):
            pb = 2 * ref.CB
            k = sum(self.m_of[points[pb * i:pb * i + pb]] * s for i, s in enumerate(ref.bytes_to_scalars(sb))) % ref.R
            want = ref.affine_to_bytes(ref.mul(k, ref.G))  # (already affine: to_affine64 below is the identity on it)
            if len(sb) // 32 <= 20000:  # ... and the C restatement of the reference's CPU MSM over Fq2 agrees (oracle/bn254.c -DORACLE_G2)
                assert self.c_model.to_affine64(self.c_model.cpu_msm(points, sb, 4)) == want, "the two G2 models disagree"
            return want

        def to_affine64(self, b):
            return b if len(b) == 2 * ref.CB else ref.affine_to_bytes(ref.jacobian_bytes_to_affine(b))

    cpu = ClosedFormG2()
elif curve != "bn254":  # another curve through the same entry points (its oracles: oracle/cpu_<curve>.py, <curve>_ref.py)
    import importlib
    cpu, ref = importlib.import_module("oracle.cpu_" + curve), importlib.import_module("oracle." + curve + "_ref")
else:
    from oracle import cpu, bn254_ref as ref
rnd = random.Random(seed)
ctx = m.MsmContext(0, curve=curve)
CBY = ctx.cb  # bytes per coordinate (48 on BLS12-381)
# BLS12-381's cofactor is not 1: the samplers' points are outside the order-r subgroup, where the endomorphism modes are not exact (and the
# R = 2^256 input format is the 4-limb curves')
MODES = ["host", "device", "windows", "batch", "group", "hostbatch", "mont", "bits", "tables", "tables_batch", "hostpipe", "mgpu", "endo", "endo_batch",
         "group_halves", "mgpu_batch", "mgpu_batch_endo", "auto", "wide", "wide_batch", "wide_shares", "mgpu_wide", "oneshot_parts", "run_parts"]
if curve == "bls12_381":
    MODES = [x for x in MODES if x not in ("mont", "endo", "endo_batch", "group_halves", "mgpu_batch_endo")]
if curve.endswith("_g2"):  # (the "mont" case below writes G1 coordinates; the pool's points are multiples of G2's generator: the endomorphism modes are exact)
    MODES = [x for x in MODES if x != "mont"]
if os.environ.get("FUZZ_MODES"):  # restrict the soak to some modes, e.g. FUZZ_MODES=wide,wide_batch
    MODES = [x for x in MODES if x in os.environ["FUZZ_MODES"].split(",")]
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
    mode = rnd.choice(MODES)
    # ("auto": the C ABI's flags = 0 -- the curve's fastest mode on a curve of prime order, the plain shape otherwise)
    wide_bits = 0
    if mode.startswith("wide"):  # every digit width the curve's scalar field admits (0: by the number of bases); BLS12-381 cannot hold 15 x 17 bits
        wide_bits = rnd.choice([0, 16, 17, 18, 19, 20] if curve not in ("bls12_381", "bls12_381_g2") else [0, 16, 18, 19, 20])
        ctx.set_wide_bits(wide_bits)
    ctx.set_bases(points, precompute="wide" if mode.startswith("wide") else mode.startswith("tables"), endomorphism=None if mode == "auto" else (mode.startswith("endo") or mode == "group_halves"))
    if mode.startswith("wide"):
        ctx.set_wide_bits(0)
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
    elif mode in ("host", "tables", "auto", "wide"):
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
    elif mode in ("tables_batch", "wide_batch"):
        k = rnd.randrange(1, 7)
        got = ctx.msm_batch(sb * k, n)[k - 1]
    elif mode == "wide_shares":
        # shares of the wide tables' virtual windows (round 5): every rank's launch of a group, what the all-gather would deliver (pairs of
        # records per virtual window) finished by msm_hip_combine_vwindows_batch_curve
        nwin = ctx.virtual_windows()
        world = rnd.choice([1, 2, 3, 5, 8, 16])
        per = -(-nwin // world)
        g = rnd.randrange(1, max(1, min(8, 64 // per)) + 1)
        pos = rnd.randrange(g)
        vecs = [bytes(len(sb)) if rnd.random() < 0.5 else sb for _ in range(g)]
        vecs[pos] = sb
        t = torch.frombuffer(bytearray(b"".join(vecs)), dtype=torch.uint8).cuda()
        pairs = []
        for r in range(world):
            b, e = window_range(r, world, nwin)
            if e == b:
                continue
            out = torch.empty((g * (e - b) * 2, 3 * CBY), dtype=torch.uint8, device="cuda")
            ctx.launch_vwindows_batch(t, n, b, e, r % 4, out)
            ctx.slot_sync(r % 4)
            pairs.append(out[pos * (e - b) * 2:(pos + 1) * (e - b) * 2])
        got = m.MsmContext.combine_vwindows_batch(torch.cat(pairs, dim=0).cpu(), nwin, curve)[0]
    elif mode == "mgpu_wide":
        # the same through msm_hip_mgpu_* (several contexts on this GPU, pinned-buffer gather), grouped launches and the synchronous call
        world = rnd.choice([1, 2, 3, 8])
        key = (world, "w")
        if key not in mg:
            mg[key] = m.MultiGpuMsm([0] * world, "host", curve=curve)
        mg[key].set_wide_bits(rnd.choice([0, 16, 18, 19, 20] if curve in ("bls12_381", "bls12_381_g2") else [0, 16, 17, 18, 19, 20]))
        mg[key].set_bases(points, precompute="wide")
        if rnd.random() < 0.5:
            got = mg[key].msm(sb)
        else:
            g = rnd.randrange(1, min(4, mg[key].group_size) + 1)
            pos = rnd.randrange(g)
            vecs = [bytes(len(sb)) if rnd.random() < 0.5 else sb for _ in range(g)]
            vecs[pos] = sb
            mg[key].launch_batch(b"".join(vecs), n, 1)
            got = mg[key].finish_batch(1, g)[pos]
    elif mode in ("oneshot_parts", "run_parts"):
        # the upload-bound call shapes as sums of sub-MSMs over ranges of the points (round 5): the one-shot call (a kept context per part) and
        # msm_hip_run with host scalars (a result slot per part, launches on ranges of the resident bases); the test hook sets the part count
        import ctypes
        parts = rnd.randrange(1, 5)
        assert m.lib().msm_hip_test_oneshot_parts(parts, 1) == 0
        try:
            if mode == "oneshot_parts":
                out = ctypes.create_string_buffer(3 * CBY)
                rc = m.lib().msm_hip_msm_curve(ctx.curve_id, points, sb, n, out)
                assert rc == 0, rc
                got = m.G1(out.raw, ctx.modulus)
            else:
                if rnd.random() < 0.5:
                    ctx.set_bases(points, endomorphism=False)
                got = ctx.msm(sb)
        finally:
            m.lib().msm_hip_test_oneshot_parts(0, 0)
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
    elif mode in ("mgpu_batch", "mgpu_batch_endo"):
        # the asynchronous grouped form (msm_hip_mgpu_launch_batch_* / finish_batch): this case's vector at a random position of a group,
        # two slots in flight, host or device-resident scalars; with endomorphism bases the devices share the 8 half-length windows
        world = rnd.choice([1, 2, 3, 5, 8, 16])
        key = (world, "b")
        if key not in mg:
            mg[key] = m.MultiGpuMsm([0] * world, "host", curve=curve)
        mg[key].set_bases(points, endomorphism=mode == "mgpu_batch_endo")
        g = rnd.randrange(1, mg[key].group_size + 1)
        pos = rnd.randrange(g)
        vecs = [bytes(len(sb)) if rnd.random() < 0.5 else sb for _ in range(g)]
        vecs[pos] = sb
        blob = b"".join(vecs)
        if rnd.random() < 0.5:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            mg[key].launch_batch([t] * world, n, 1)
            mg[key].launch_batch([t] * world, n, 2)
        else:
            mg[key].launch_batch(blob, n, 1)
            mg[key].launch_batch(blob, n, 2)
        a, b2 = mg[key].finish_batch(1, g), mg[key].finish_batch(2, g)
        assert a[pos] == b2[pos]
        got = a[pos]
    elif mode == "group_halves":
        # grouped half-length window shares (endomorphism bases): every rank's share of a group, gathered and combined (8 sums per MSM)
        world = rnd.choice([2, 3, 4, 8])
        per = -(-8 // world)
        g = max(1, 8 // per)
        pos = rnd.randrange(g)
        vecs = [bytes(len(sb)) if rnd.random() < 0.5 else sb for _ in range(g)]
        vecs[pos] = sb
        t = torch.frombuffer(bytearray(b"".join(vecs)), dtype=torch.uint8).cuda()
        parts = []
        for r in range(world):
            b, e = window_range(r, world, 8)
            out = torch.empty((g * (e - b), 3 * CBY), dtype=torch.uint8, device="cuda")
            ctx.launch_half_windows_batch(t, n, b, e, r % 4, out)
            ctx.slot_sync(r % 4)
            parts.append(out[pos * (e - b):(pos + 1) * (e - b)])
        got = combine(torch.cat(parts, dim=0))
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
            out = torch.empty((g * (e - b), 3 * CBY), dtype=torch.uint8, device="cuda")
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
m.lib().msm_hip_oneshot_release()  # (the contexts the one-shot cases left behind)
print("fuzz ok: %d cases in %.1f s (seed %d, %s)" % (cases, time.time() - t0, seed, curve))
