"""G2 behind the same ABI and kernels (SURVEY.md 8f-4 "other curves / G2"; the reference lists other curves as future work, README.md, and
is hard-wired to BN254 G1, src/cuzk/msm.rs:37-43): BN254's twist y^2 = x^3 + 3 / (9 + u) and BLS12-381's y^2 = x^3 + 4 (1 + u), both over
Fq2 = Fq[u] / (u^2 + 1), scalars modulo the curve's r.  The coordinates are Fq2 elements -- csrc/fq2.h on the curve's prime-field unit,
18 (BN254) or 28 (BLS12-381) limbs per coordinate -- c0 || c1 on the wire: coordinates 64 / 96 B, points 128 / 192 B, Jacobian records
192 / 288 B (csrc/curve_bn254_g2.hip, csrc/curve_bls12_381_g2.hip).

Checked against BOTH models of each curve's G2 (pinned and cross-checked in tests/test_oracle_g2.py): the pure-Python one
(oracle/bn254_g2_ref.py / bls12_381_g2_ref.py: Pippenger and double-and-add MSMs at the sizes Python finishes in seconds) and the G2 builds of
the C restatement (oracle/bn254.c -DORACLE_G2: the reference's CPU MSM and its stage models over Fq2, at full size) -- and through a closed
form the synthetic points offer: they are KNOWN multiples m_i G of the generator, so sum_i s_i P_i = (sum_i s_i m_i mod r) G whatever the
size and whatever method anybody used."""
import os
import importlib

import pytest
import torch

import msm_webgpu_amd as m
from tests.util import rng

pytestmark = pytest.mark.gpu
CURVE_IDS = {"bn254_g2": 5, "bls12_381_g2": 6}


def b32(x):
    return int(x).to_bytes(32, "little")


@pytest.fixture(scope="module", params=["bn254_g2", "bls12_381_g2"])
def env(request, built):
    """The context of the curve under test; the module's globals g2 / cpu (its Python / C model), FB, CB, PB, JB follow it"""
    global g2, cpu, FB, CB, PB, JB
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g2 = importlib.import_module("oracle." + request.param + "_ref")
    cpu = importlib.import_module("oracle.cpu_" + request.param)
    FB, CB, PB, JB = g2.FB, g2.CB, 2 * g2.CB, 3 * g2.CB
    c = m.MsmContext(0, curve=request.param)
    yield c
    c.close()


@pytest.fixture
def ctx(env):
    return env


def bf(x):
    return int(x).to_bytes(FB, "little")


def affs(raw):
    return [g2.jacobian_bytes_to_affine(raw[i:i + JB]) for i in range(0, len(raw), JB)]


def jac(pt, r):
    """A Jacobian record of `pt` with a random z"""
    if pt is None:
        return bytes(JB)
    z = (r.randrange(1, g2.P), r.randrange(g2.P))
    z2 = g2.f2_sqr(z)
    return g2.f2_to_bytes(g2.f2_mul(pt[0], z2)) + g2.f2_to_bytes(g2.f2_mul(pt[1], g2.f2_mul(z2, z))) + g2.f2_to_bytes(z)


@pytest.mark.parametrize("op", ["add", "sub", "mul", "sqr", "neg", "mul_asm", "sqr_asm", "mul2_asm", "mul_asm_lazy", "sqr_asm_lazy"])
def test_field_ops(ctx, op):
    # (≙ tests/field.rs) Fq2 on the device: the C++ form of the prime-field multipliers and the SMVP's inline-assembly form
    P = g2.P
    r = rng(61)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << (P.bit_length() - 1), (1 << 29) - 1, 1 << 28]
    vals = [(a, b) for a in edge for b in edge] + [(r.randrange(P), r.randrange(P)) for _ in range(3000)]
    other = [vals[(7 * i + 3) % len(vals)] for i in range(len(vals))]
    a, b = b"".join(g2.f2_to_bytes(v) for v in vals), b"".join(g2.f2_to_bytes(v) for v in other)
    mul, add = g2.f2_mul, g2.f2_add
    want = {"add": add, "sub": g2.f2_sub, "mul": mul, "sqr": lambda x, y: mul(x, x), "neg": lambda x, y: g2.f2_neg(x), "mul_asm": mul,
            "sqr_asm": lambda x, y: mul(x, x), "mul2_asm": lambda x, y: add(mul(x, y), mul(y, x)), "mul_asm_lazy": lambda x, y: mul(add(x, y), add(x, x)),
            "sqr_asm_lazy": lambda x, y: mul(add(x, y), add(x, y))}[op]
    assert ctx.fq_op(op, a, b) == b"".join(g2.f2_to_bytes(want(x, y)) for x, y in zip(vals, other))


def test_point_ops_and_special_cases(ctx):
    # (≙ tests/point.rs) the group formulas over Fq2 with the reference's case split: identity operands, P + P, P - P
    r = rng(62)
    pts = g2.sample_points(33, 12)
    a = pts[:6] + [None, pts[3], pts[4], pts[5], None]
    b = pts[6:12] + [pts[2], None, pts[4], g2.neg(pts[5]), None]
    A, B = b"".join(jac(x, r) for x in a), b"".join(jac(x, r) for x in b)
    assert affs(ctx.g1_op("add", A, B)) == [g2.add(x, y) for x, y in zip(a, b)]
    assert affs(ctx.g1_op("double", A)) == [g2.add(x, x) for x in a]
    q = pts[1:7] + [pts[6], pts[3], g2.neg(pts[4]), pts[5], pts[7]]
    Q = g2.points_to_bytes(q)
    assert affs(ctx.g1_op("add_affine", A, Q)) == [g2.add(x, y) for x, y in zip(a, q)]
    assert affs(ctx.g1_op("madd_w_pmp", A, Q)) == [g2.add(x, y) for x, y in zip(a, q)]
    assert affs(ctx.g1_op("madd_w_mm", A, Q)) == [g2.add(g2.add(x, g2.neg(y)), g2.neg(y)) for x, y in zip(a, q)]
    ks = [0, 1, 2, 3, 0xFFFF, 0x8000, 0xFFFFFFFF, 12345, 7, 1 << 31, 99]
    assert affs(ctx.g1_mul_u32(A, ks)) == [g2.mul(k, x) for k, x in zip(ks, a)]


def test_golden_vectors(ctx):
    # the committed known answers (tests/golden/msm_vectors_g2.json): explicit edge cases (2 G, (r - 1) G, cancellations, the 0x8000 digit
    # chain, duplicates and P / -P in one bucket, all-equal scalars) and seeded sizes up to 2^16 + 4
    from tests.util import case_inputs_g2, golden_cases_g2

    cases = [c for c in golden_cases_g2() if c["curve"] == ctx.curve]
    assert len(cases) == 14
    for case in cases:
        pb, sb = case_inputs_g2(case)
        ctx.set_bases(pb, check_on_curve=True)
        assert ctx.msm(sb).to_affine_bytes() == bytes.fromhex(case["expected_affine"]), case["name"]


@pytest.mark.parametrize("n", [1, 3, 257, 4097])
def test_msm_matches_the_model_every_window_size_and_entry_point(ctx, n):
    pts = g2.sample_points(n, 34)
    sc = [g2.sample_scalar(35, i) for i in range(n)]
    for i, v in enumerate([0, 1, g2.R - 1, g2.R - 2, 0x8000, (1 << 253) + 0x80008000][: min(n, 6)]):
        sc[i] = v
    if n > 40:  # a duplicate and a negated duplicate of a point with equal scalars: the mixed addition's doubling / cancellation paths
        pts[20], sc[20] = pts[21], sc[21]
        pts[23], sc[23] = g2.neg(pts[22]), sc[22]
    want = g2.msm_pippenger(pts, sc, c=10)
    if n <= 257:
        assert want == g2.msm_naive(pts, sc)
    points, scb = g2.points_to_bytes(pts), g2.scalars_to_bytes(sc)
    dev = torch.frombuffer(bytearray(scb), dtype=torch.uint8).cuda()
    ctx.set_bases(points, check_on_curve=True)
    for bits in (0, 12, 14, 16):
        ctx.set_window_bits(bits)
        got = ctx.msm(scb)
        assert len(got.xyz) == JB and got.to_affine() == want and got.to_affine_bytes() == g2.affine_to_bytes(want), (n, bits)
    ctx.set_window_bits(0)
    assert ctx.msm(dev).to_affine() == want
    assert [g.to_affine() for g in ctx.msm_batch(scb * 3, n)] == [want] * 3
    ctx.launch(dev, 0)
    ctx.launch(dev, 1)
    assert ctx.finish(1).to_affine() == want and ctx.finish(0).to_affine() == want
    # window shards combine to the whole (the multi-GPU decomposition); grouped shards; the in-process multi-GPU ABI
    parts = [ctx.msm_windows(dev, 0, 6), ctx.msm_windows(dev, 6, 16)]
    assert parts[0].shape == (6, JB)
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0), curve=ctx.curve).to_affine() == want
    two = torch.cat([dev, dev])
    out = torch.zeros((2 * 4, JB), dtype=torch.uint8, device=dev.device)
    shards = []
    for k in range(4):
        ctx.launch_windows_batch(two, n, 4 * k, 4 * k + 4, k % 3, out)
        ctx.slot_sync(k % 3)
        shards.append(out.clone().view(2, 4, JB))
    for v in range(2):
        assert m.MsmContext.combine_windows(torch.cat([s[v] for s in shards], dim=0), curve=ctx.curve).to_affine() == want
    mg = m.MultiGpuMsm([0, 0, 0], "host", curve=ctx.curve)
    try:
        mg.set_bases(points)
        assert mg.msm(scb).to_affine() == want
        mg.launch_batch(scb * 2, n, 1)
        assert [g.to_affine() for g in mg.finish_batch(1, 2)] == [want] * 2
    finally:
        mg.close()
    # fixed-base tables (one bucket set for all windows; an inversion in Fq2 per table entry), whole MSMs and batches
    ctx.set_bases(points, precompute=True)
    assert ctx.msm(scb).to_affine() == want and ctx.msm(dev).to_affine() == want
    assert [g.to_affine() for g in ctx.msm_batch(scb * 2, n)] == [want] * 2
    # ... and the wide tables (14 digits of 19 bits into one bucket set of 2^18 slots)
    ctx.set_bases(points, precompute="wide")
    assert ctx.msm(scb).to_affine() == want and ctx.msm(dev).to_affine() == want
    assert [g.to_affine() for g in ctx.msm_batch(scb * 2, n)] == [want] * 2
    # scalars handed over as s * 2^256 mod r (the in-memory words of a 4 x 64-bit Montgomery library)
    ctx.set_bases(points)
    ctx.set_scalar_format(True)
    try:
        assert ctx.msm(b"".join(b32((v << 256) % g2.R) for v in sc)).to_affine() == want
    finally:
        ctx.set_scalar_format(False)


def test_bucket_sums_and_window_sums_against_the_model(ctx):
    # (≙ tests/smvp_shader.rs:292-334, tests/cuzk.rs) stage read-back: every bucket of two windows is the signed sum of the points whose
    # digit selects it (digits: the reference's recode, decompose_scalars.template.wgsl:83-112); the window sums are sum_k k B_k
    n = 3000
    pts = g2.sample_points(n, 44)
    sc = [g2.sample_scalar(45, i) for i in range(n)]
    ctx.set_bases(g2.points_to_bytes(pts))
    ctx.set_debug(True)
    ctx.set_window_bits(16)
    try:
        result = ctx.msm(g2.scalars_to_bytes(sc))
    finally:
        ctx.set_debug(False)
        ctx.set_window_bits(0)
    buckets, wsums = ctx.read_buckets(16, 1 << 15), ctx.read_window_sums(16)
    digs = [g2.signed_digits(s) for s in sc]
    for w in (0, 15):
        want = {}
        for i in range(n):
            d = digs[i][w]
            if d:
                k = abs(d) & 0x7FFF  # slot 0 holds the digit -2^15
                want[k] = g2.add(want.get(k), pts[i] if d > 0 else g2.neg(pts[i]))
        raw = buckets[w].tobytes()
        occupied = {k: g2.jacobian_bytes_to_affine(raw[JB * k:JB * k + JB]) for k in range(1 << 15) if raw[JB * k + 2 * CB:JB * k + JB] != bytes(CB)}
        assert occupied == {k: v for k, v in want.items() if v is not None}
        total = None
        for k, v in want.items():
            total = g2.add(total, g2.mul(k if k else 1 << 15, v))
        assert g2.jacobian_bytes_to_affine(wsums[w].tobytes()) == total
    acc = None
    for w in range(15, -1, -1):
        for _ in range(16):
            acc = g2.add(acc, acc)
        acc = g2.add(acc, g2.jacobian_bytes_to_affine(wsums[w].tobytes()))
    assert acc == result.to_affine() == g2.msm_by_multipliers(g2.sample_multipliers(n, 44), sc)
    # without debug the single-MSM launch leaves bit-plane sums: the read-back finishes them
    ctx.set_window_bits(16)
    try:
        ctx.msm(g2.scalars_to_bytes(sc))
        assert m.MsmContext.combine_windows(ctx.read_window_sums(16).tobytes(), curve=ctx.curve).to_affine() == acc
    finally:
        ctx.set_window_bits(0)


@pytest.mark.parametrize("logn", [16, 18, 20])
def test_large_msm_against_the_cpu_msm_and_the_closed_form(ctx, logn):
    # BASELINE config sizes (2^16: config 1; 2^20: config 2) on G2: 2^logn DISTINCT points (the C model's sampler: the same known multiples of
    # the generator as the Python model's), the result against the C restatement of the reference's CPU MSM on the same inputs AND against the
    # closed form (sum_i s_i m_i mod r) G
    if logn == 20 and ctx.curve != "bn254_g2":
        pytest.skip("2^20 on the reference's curve only (2^18 covers the 28-limb unit)")
    n = 1 << logn
    pb = cpu.sample_points(70 + logn, n)
    assert pb[: 8 * PB] == g2.points_to_bytes(g2.sample_points(8, 70 + logn))
    ms = g2.sample_multipliers(n, 70 + logn)
    ctx.set_bases(pb, check_on_curve=True)
    sc = ctx.sample_scalars(n, 71 + logn)
    sb = sc.cpu().numpy().tobytes()
    assert sb[: 32 * 100] == cpu.sample_scalars(71 + logn, 100)
    want = g2.msm_by_multipliers(ms, g2.bytes_to_scalars(sb))
    assert cpu.to_affine64(cpu.cpu_msm(pb, sb, max(1, min(16, os.cpu_count() or 1)))) == g2.affine_to_bytes(want)
    assert ctx.msm(sc).to_affine() == want
    ctx.launch(sc, 0)
    ctx.launch(sc, 1)
    assert ctx.finish(0).to_affine() == want and ctx.finish(1).to_affine() == want
    if logn == 16:
        # skew: every scalar equal; a witness-like vector (70 % zeros and ones); one rank's shares of an 8-rank run, gathered in rank order
        s = (0x1234_5678_9ABC_DEF0_1357_9BDF_2468_ACE0_FEDC_BA98_7654_3210 * 0x10001) % g2.R
        assert ctx.msm(b32(s) * n).to_affine() == g2.mul(s * sum(ms), g2.G)
        gen = torch.Generator(device=sc.device)
        gen.manual_seed(3)
        sel = torch.rand(n, device=sc.device, generator=gen)
        wit = sc.clone()
        wit[sel < 0.7] = 0
        wit[(sel >= 0.4) & (sel < 0.7), 0] = 1
        wb = wit.cpu().numpy().tobytes()
        assert ctx.msm(wit).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, wb, 8)) == g2.affine_to_bytes(g2.msm_by_multipliers(ms, g2.bytes_to_scalars(wb)))
        out = torch.zeros((16, JB), dtype=torch.uint8, device=sc.device)
        for rank in range(8):
            ctx.launch_windows_batch(sc, n, 2 * rank, 2 * rank + 2, rank % 3, out[2 * rank:2 * rank + 2])
            ctx.slot_sync(rank % 3)
        assert m.MsmContext.combine_windows(out, curve=ctx.curve).to_affine() == want
    ctx.set_bases(pb[: 4 * PB])


def test_stages_against_the_cpu_stage_models(ctx):
    # (≙ tests/smvp_shader.rs:292-334, tests/cuzk.rs) bucket sums and window sums of the device against the C restatement of the reference's CPU
    # models (src/cuzk/test/utils.rs:61-338) over Fq2
    n = 20000
    points, scalars = cpu.sample_points(60, n), cpu.sample_scalars(61, n)
    ctx.set_bases(points)
    ctx.set_debug(True)
    ctx.set_window_bits(16)
    try:
        result = ctx.msm(scalars)
    finally:
        ctx.set_debug(False)
        ctx.set_window_bits(0)
    buckets, wsums = ctx.read_buckets(16, 1 << 15), ctx.read_window_sums(16)
    digits = cpu.decompose_scalars_signed(scalars, 16, 16)
    for w in (3, 15):
        cp, vi = cpu.transpose(digits[w], 1 << 16)
        want = cpu.smvp_signed(cp, vi, points, 1 << 16)
        got = buckets[w].tobytes()
        assert [cpu.to_affine64(got[i:i + JB]) for i in range(0, len(got), JB)] == [cpu.to_affine64(want[i:i + JB]) for i in range(0, len(want), JB)]
        assert cpu.to_affine64(wsums[w].tobytes()) == cpu.to_affine64(cpu.bucket_reduction("running_sum", got))
    assert cpu.to_affine64(cpu.horner(wsums.tobytes(), 16)) == result.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, scalars))
    assert result.to_affine_bytes() == cpu.to_affine64(cpu.msm_cuzk_model(points, scalars))


def test_window_sharded_pipeline_on_this_curve(ctx):
    # msm-webgpu_amd/sharding.py (the one-process-per-GPU path of bench.py) with this curve's record size: the whole-MSM helper and the
    # grouped asynchronous pipeline with one rank, and one rank's partial sums of an 8-rank run against the other seven's from msm_windows
    from msm_webgpu_amd.sharding import ShardedMsmPipeline, sharded_msm, window_range

    n = 3000
    pb, sb = cpu.sample_points(81, n), cpu.sample_scalars(82, n)
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb))
    ctx.set_bases(pb)
    dev = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
    assert sharded_msm(ctx, dev, 0, 1).to_affine_bytes() == want
    pipe = ShardedMsmPipeline(ctx, 0, 1, msms_per_issue=2)
    two = torch.cat([dev, dev]).contiguous()
    torch.cuda.synchronize()
    pipe.issue(two, n=n, inputs_complete=True)
    pipe.issue(two, n=n, inputs_complete=True)
    assert [g.to_affine_bytes() for g in pipe.complete() + pipe.complete()] == [want] * 4
    emu = ShardedMsmPipeline(ctx, 0, 1, msms_per_issue=1, emulate_world=8)  # rank 0's two windows of an 8-rank run (partial sums)
    emu.issue(dev, inputs_complete=True)
    part0 = emu.complete()
    b, e = window_range(0, 8)
    rest = ctx.msm_windows(dev, e, 16)
    mine = ctx.msm_windows(dev, b, e)
    assert part0.to_affine_bytes() == m.MsmContext.combine_windows(mine, curve=ctx.curve).to_affine_bytes()
    assert m.MsmContext.combine_windows(torch.cat([mine, rest], dim=0), curve=ctx.curve).to_affine_bytes() == want


@pytest.mark.parametrize("n", [1, 2, 65, 257, 2049])
def test_endomorphism_mode_on_g2(ctx, n):
    """MSM_HIP_BASES_ENDOMORPHISM on G2 (round 4): the twist has j = 0, so phi(x, y) = (beta x, y), beta a cube root of unity of the prime field,
    is the multiplication by lambda on the order-r subgroup; every scalar is split k = k1 + k2 lambda on the device (the G1 unit's split) and
    the MSM runs over the 2n points P_i, phi(P_i) with half the windows.  Against the model's plain MSM, for every window size and entry point,
    with the adversarial scalars of the G1 test; the model's own phi and split are checked against lambda P first."""
    q = g2.glv_params()
    lam = q["lam"]
    pts = g2.sample_points(n, 44)
    assert all(g2.endo(pt) == g2.mul(lam, pt) for pt in pts[:3])
    sc = [g2.sample_scalar(45, i) for i in range(n)]
    special = [0, 1, g2.R - 1, lam, g2.R - lam, 0x8000, (0x8000 * lam) % g2.R, g2.R - 0x8000, (1 << 253) + 0x80008000]
    for i, v in enumerate(special[: n]):
        sc[i] = v
    for k in sc[:12]:
        k1, k2 = g2.glv_split(k)
        assert (k1 + k2 * lam) % g2.R == k % g2.R and abs(k1) < 1 << 127 and abs(k2) < 1 << 127
    if n > 40:  # duplicates / negated duplicates: P_i meets itself (or phi(P_i) does) in a bucket
        pts[20], sc[20] = pts[21], sc[21]
        pts[23], sc[23] = g2.neg(pts[22]), sc[22]
    want = g2.msm_pippenger(pts, sc, c=10)
    points, scb = g2.points_to_bytes(pts), g2.scalars_to_bytes(sc)
    dev = torch.frombuffer(bytearray(scb), dtype=torch.uint8).cuda()
    ctx.set_bases(points, check_on_curve=True, endomorphism=True)
    assert ctx.uses_endomorphism()
    try:
        for bits in (0, 12, 14, 16):
            ctx.set_window_bits(bits)
            assert ctx.msm(scb).to_affine() == want, (n, bits)
        ctx.set_window_bits(0)
        assert ctx.msm(dev).to_affine() == want
        assert [g.to_affine() for g in ctx.msm_batch(scb * 3, n)] == [want] * 3
        ctx.launch(dev, 0)
        ctx.launch_host(scb, 1)
        assert ctx.finish(1).to_affine() == want and ctx.finish(0).to_affine() == want
        # the plain window shards still work on the same base set (records 0 .. n - 1), and so do the half-length shards
        parts = [ctx.msm_windows(dev, 0, 6), ctx.msm_windows(dev, 6, 16)]
        assert m.MsmContext.combine_windows(torch.cat(parts, dim=0), curve=ctx.curve).to_affine() == want
        mg = m.MultiGpuMsm([0, 0, 0], "host", curve=ctx.curve)
        try:
            mg.set_bases(points, endomorphism=True)
            assert mg.msm(scb).to_affine() == want
            mg.launch_batch(scb * 2, n, 1)
            assert [g.to_affine() for g in mg.finish_batch(1, 2)] == [want] * 2
        finally:
            mg.close()
    finally:
        ctx.set_window_bits(0)
        ctx.set_bases(points)
    assert not ctx.uses_endomorphism() and ctx.msm(scb).to_affine() == want


def test_options_and_input_errors_of_the_g2_unit(ctx):
    pts = g2.points_to_bytes(g2.sample_points(4, 38))
    with pytest.raises(m.MsmHipError) as e:  # the endomorphism images and the fixed-base tables exclude each other, as on G1
        ctx.set_bases(pts, endomorphism=True, precompute=True)
    assert e.value.code == -2
    ctx.set_bases(pts, endomorphism=None)  # the ABI's default (flags = 0) on a curve with a cofactor: the plain shape
    assert not ctx.uses_endomorphism()
    # the device point sampler (round 4): P_i = (a + i b) G, byte for byte the model's sample_points -- points of G2 proper
    for n, seed in ((1, 3), (70, 77)):
        assert ctx.sample_points(n, seed).cpu().numpy().tobytes() == g2.points_to_bytes(g2.sample_points(n, seed))
    with pytest.raises(m.MsmHipError):  # a component >= p (c1 of x)
        ctx.set_bases(pts[:FB] + bf(g2.P) + pts[CB:])
    with pytest.raises(m.MsmHipError):  # not on the twist
        ctx.set_bases(bf(5) + bf(6) + bf(7) + bf(8) + pts[PB:], check_on_curve=True)
    ctx.set_bases(pts, check_on_curve=True)
    with pytest.raises(m.MsmHipError):  # a scalar that overflows the 16-bit recode
        ctx.msm(b"\xff" * 32 + bytes(96))
    assert ctx.msm(bytes(128)).is_identity()
    assert m.lib().msm_hip_ctx_curve(ctx._h) == CURVE_IDS[ctx.curve]
    # Montgomery-form coordinates (x * 2^256 mod p per component; 2^384 on BLS12-381: the host library's radix) are accepted: the same result
    mont = b"".join(bf(int.from_bytes(pts[i:i + FB], "little") * (1 << (8 * FB)) % g2.P) for i in range(0, len(pts), FB))
    sc = g2.scalars_to_bytes([5, 6, 7, 8])
    want = ctx.msm(sc).to_affine()
    ctx.set_bases(mont, mont256=True)
    assert ctx.msm(sc).to_affine() == want == g2.msm_naive(g2.bytes_to_points(pts), [5, 6, 7, 8])
