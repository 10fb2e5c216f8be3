"""The Pasta cycle behind the same ABI and kernels (SURVEY.md 8f-4 "other curves (Pallas/Vesta ...)"; the reference's own dead second
curve is Pallas, src/naive/wgsl/pallas, and its README lists other curves as future work): y^2 = x^3 + 5 over the 255-bit p (Pallas, q
points) and over q (Vesta, p points).  Their moduli leave 2^261 / p = 127 of Montgomery headroom instead of BN254's 169 -- the group
formulas need 121.5 -- so these tests also stand for "the arithmetic does not depend on BN254's modulus".
Checked against the Pasta builds of both oracles (oracle/bn254.c -DORACLE_PALLAS / -DORACLE_VESTA, oracle/{pallas,vesta}_ref.py)."""
import importlib

import pytest
import torch

import msm_webgpu_amd as m
from tests.util import rng

pytestmark = pytest.mark.gpu


def b32(x):
    return int(x).to_bytes(32, "little")


@pytest.fixture(scope="module", params=["pallas", "vesta"])
def cv(built, request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    name = request.param
    cpu = importlib.import_module("oracle.cpu_" + name)
    ref = importlib.import_module("oracle." + name + "_ref")
    c = m.MsmContext(0, curve=name)
    yield name, c, cpu, ref
    c.close()


def test_oracles_agree_and_know_the_generator(cv):
    name, ctx, cpu, ref = cv
    assert ref.G == (ref.P - 1, 2) and ref.mul(ref.R, ref.G) is None  # (-1, 2) generates a group of order r
    pts, sc = ref.sample_points(3, 7), ref.sample_scalars(4, 7)
    assert cpu.sample_points(3, 7) == ref.points_to_bytes(pts) and cpu.sample_scalars(4, 7) == ref.scalars_to_bytes(sc)
    want = ref.affine_to_bytes64(ref.msm_naive(pts, sc))
    assert cpu.to_affine64(cpu.cpu_msm(ref.points_to_bytes(pts), ref.scalars_to_bytes(sc))) == want


@pytest.mark.parametrize("op", ["add", "sub", "mul", "sqr", "neg", "mul_asm", "sqr_asm", "mul2_asm", "mul_asm_lazy", "sqr_asm_lazy"])
def test_field_ops(cv, op):
    name, ctx, cpu, ref = cv
    P = ref.P
    r = rng(31)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 254, (1 << 254) - 1, (1 << 29) - 1, 1 << 29, 0x1FFFFFFF << 29, (1 << 64) - 1]
    vals = edge + [r.randrange(P) for _ in range(3000)]
    a = b"".join(b32(v) for v in vals)
    b = b"".join(b32(vals[(7 * i + 3) % len(vals)]) for i in range(len(vals)))
    mul, add = (lambda x, y: cpu.fq_op("mul", x, y)), (lambda x, y: cpu.fq_op("add", x, y))
    want = {"mul_asm": lambda: mul(a, b), "sqr_asm": lambda: mul(a, a), "mul2_asm": lambda: add(mul(a, b), mul(b, a)),
            "mul_asm_lazy": lambda: mul(add(a, b), add(a, a)), "sqr_asm_lazy": lambda: mul(add(a, b), add(a, b))}
    assert ctx.fq_op(op, a, b) == (want[op]() if op in want else cpu.fq_op(op, a, b))


def test_point_ops_and_special_cases(cv):
    name, ctx, cpu, ref = cv
    r = rng(32)
    pts = ref.sample_points(33, 12)

    def jac(pt):
        if pt is None:
            return bytes(96)
        z = r.randrange(1, ref.P)
        return b32(pt[0] * z * z % ref.P) + b32(pt[1] * z * z * z % ref.P) + b32(z)

    def aff64(raw):
        return [cpu.to_affine64(raw[i:i + 96]) for i in range(0, len(raw), 96)]

    a = pts[:6] + [None, pts[3], pts[4], pts[5], None]
    b = pts[6:12] + [pts[2], None, pts[4], ref.neg(pts[5]), None]
    A, B = b"".join(jac(x) for x in a), b"".join(jac(x) for x in b)
    assert aff64(ctx.g1_op("add", A, B)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, b)]
    assert aff64(ctx.g1_op("double", A)) == [ref.affine_to_bytes64(ref.add(x, x)) for x in a]
    q = pts[1:7] + [pts[6], pts[3], ref.neg(pts[4]), pts[5], pts[7]]
    Q = ref.points_to_bytes(q)
    assert aff64(ctx.g1_op("add_affine", A, Q)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert aff64(ctx.g1_op("madd_w_pmp", A, Q)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert aff64(ctx.g1_op("madd_w_mm", A, Q)) == [ref.affine_to_bytes64(ref.add(ref.add(x, ref.neg(y)), ref.neg(y))) for x, y in zip(a, q)]


@pytest.mark.parametrize("n", [1, 3, 257, 4097, 70001])
def test_msm_matches_oracle_every_window_size_and_mode(cv, n):
    name, ctx, cpu, ref = cv
    points, sc = bytearray(cpu.sample_points(34, n)), bytearray(cpu.sample_scalars(35, n))
    edge = [0, 1, ref.R - 1, ref.R - 2, 0x8000, (1 << 254) - 1 if (1 << 254) - 1 < ref.R else ref.R - 3]
    for i, v in enumerate(edge[: min(n, len(edge))]):
        sc[32 * i:32 * i + 32] = b32(v)
    points, sc = bytes(points), bytes(sc)
    want = cpu.to_affine64(cpu.cpu_msm(points, sc, 8))
    dev = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    for endo in (False, True):
        ctx.set_bases(points, check_on_curve=True, endomorphism=endo)
        for bits in (0, 12, 14, 16):
            ctx.set_window_bits(bits)
            assert ctx.msm(sc).to_affine_bytes() == want, (n, endo, bits)
        ctx.set_window_bits(0)
        assert ctx.msm(dev).to_affine_bytes() == want
        assert [g.to_affine_bytes() for g in ctx.msm_batch(sc * 3, n)] == [want] * 3
    # window shards combine to the whole (the multi-GPU decomposition), fixed-base tables, the in-process multi-GPU ABI
    ctx.set_bases(points)
    parts = [ctx.msm_windows(dev, 0, 6), ctx.msm_windows(dev, 6, 16)]
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0), curve=name).to_affine_bytes() == want
    if n <= 4097:
        ctx.set_bases(points, precompute=True)
        assert ctx.msm(sc).to_affine_bytes() == want
        ctx.set_bases(points, precompute="wide")          # 14 digits of 19 bits (a 255-bit modulus: the top digit reaches 128), 8 virtual windows
        assert ctx.msm(sc).to_affine_bytes() == want and ctx.msm(dev).to_affine_bytes() == want
        assert ctx.wide_bits() == 16                      # (the policy's width for a small base set)
        # 17 bits: the moduli are 2^254 + a 126-bit number -- the top digit of 15 x 17 bits reaches exactly 2^16 for these:
        top = [ref.R - 1, ref.R - 2, 1 << 254, (1 << 254) + 12345, (1 << 254) - 1][: min(n, 5)]
        tb = ref.scalars_to_bytes(top)
        assert ctx.msm(tb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[: 64 * len(top)], tb, 1))
        for bits in (17, 19, 20):
            ctx.set_wide_bits(bits)
            try:
                ctx.set_bases(points, precompute="wide")
                assert ctx.wide_bits() == bits
                assert ctx.msm(sc).to_affine_bytes() == want and ctx.msm(tb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[: 64 * len(top)], tb, 1))
            finally:
                ctx.set_wide_bits(0)
        ctx.set_bases(points)
        mg = m.MultiGpuMsm([0, 0], "host", curve=name)
        try:
            mg.set_bases(points)
            assert mg.msm(sc).to_affine_bytes() == want
        finally:
            mg.close()


def test_samplers_and_large_msm(cv):
    name, ctx, cpu, ref = cv
    n = 1 << 17
    pts, sc = ctx.sample_points(n, 36), ctx.sample_scalars(n, 37)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    assert pb[: 64 * 500] == cpu.sample_points(36, 500) and sb[: 32 * 500] == cpu.sample_scalars(37, 500)
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, 8))
    for endo in (False, True):
        ctx.set_bases(pts, endomorphism=endo)
        assert ctx.msm(sc).to_affine_bytes() == want
    # skew: every scalar equal, and a witness-like vector
    s = (0x1234_5678_9ABC_DEF0_1357_9BDF_2468_ACE0_FEDC_BA98_7654_3210 * 0x10001) % ref.R
    eq = b32(s) * n
    assert ctx.msm(eq).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, eq, 8))


def test_2p20_points_bit_exact(cv):
    # BASELINE config 2's size on the other curves: plain and endomorphism bases against the multithreaded oracle
    import os

    name, ctx, cpu, ref = cv
    n = 1 << 20
    pts, sc = ctx.sample_points(n, 40), ctx.sample_scalars(n, 41)
    want = cpu.to_affine64(cpu.cpu_msm(pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes(), max(1, min(16, os.cpu_count() or 1))))
    for endo in (False, True):
        ctx.set_bases(pts, endomorphism=endo)
        assert ctx.msm(sc).to_affine_bytes() == want, endo
    ctx.set_bases(pts[:4].contiguous())


def test_input_errors(cv):
    name, ctx, cpu, ref = cv
    pts = cpu.sample_points(38, 4)
    with pytest.raises(m.MsmHipError):  # coordinate >= p
        ctx.set_bases(b32(ref.P) + pts[32:])
    with pytest.raises(m.MsmHipError):  # not on the curve
        ctx.set_bases(b32(5) + b32(7) + pts[64:], check_on_curve=True)
    ctx.set_bases(pts)
    with pytest.raises(m.MsmHipError):  # a scalar that overflows the 16-bit recode
        ctx.msm(b"\xff" * 32 + bytes(96))
    assert ctx.msm(bytes(128)).is_identity()
