"""BLS12-381 G1 behind the same ABI and kernels (SURVEY.md 8f-4 "other curves (... BLS12-381)"; the reference lists other curves as future
work, README.md, and is hard-wired to BN254's Fq, src/cuzk/msm.rs:37-43): y^2 = x^3 + 4 over the 381-bit p, scalars modulo the 255-bit r.
The field is wider than the others': 14 limbs of 28 bits on the device (csrc/curve_bls12_381.hip), 48-byte coordinates on the wire (points
96 B, Jacobian records 144 B).  Checked against the BLS12-381 builds of both oracles (oracle/bn254.c -DORACLE_BLS12_381,
oracle/bls12_381_ref.py).  The curve's cofactor is not 1: the samplers' points lie on the curve but not in the order-r subgroup, which is
fine for every mode that treats scalars as integers; the endomorphism mode (k = k1 + k2 lambda mod r) is exact for points of order r only --
what every valid G1 input is -- and is tested with multiples of the generator."""
import os

import pytest
import torch

import msm_webgpu_amd as m
from oracle import bls12_381_ref as ref
from oracle import cpu_bls12_381 as cpu
from tests.util import rng

pytestmark = pytest.mark.gpu
CB, PB, JB = 48, 96, 144


def b48(x):
    return int(x).to_bytes(48, "little")


def b32(x):
    return int(x).to_bytes(32, "little")


@pytest.fixture(scope="module")
def ctx(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = m.MsmContext(0, curve="bls12_381")
    yield c
    c.close()


def subgroup_points(seed, n):
    """n points of order r: multiples of the generator (wire format)"""
    jac = cpu.g1_scalar_mul(ref.points_to_bytes([ref.G]) * n, cpu.sample_scalars(seed, n))
    return b"".join(cpu.to_affine64(jac[JB * i:JB * (i + 1)]) for i in range(n))


def test_oracles_agree_and_know_the_generator():
    assert ref.mul(ref.R, ref.G) is None and ref.is_on_curve(ref.G)
    pts, sc = ref.sample_points(3, 7), ref.sample_scalars(4, 7)
    assert cpu.sample_points(3, 7) == ref.points_to_bytes(pts) and cpu.sample_scalars(4, 7) == ref.scalars_to_bytes(sc)
    want = ref.affine_to_bytes64(ref.msm_naive(pts, sc))
    assert cpu.to_affine64(cpu.cpu_msm(ref.points_to_bytes(pts), ref.scalars_to_bytes(sc))) == want
    assert len(want) == PB and cpu.coord_bytes() == CB


@pytest.mark.parametrize("op", ["add", "sub", "mul", "sqr", "neg", "mul_asm", "sqr_asm", "mul2_asm", "mul_asm_lazy", "sqr_asm_lazy"])
def test_field_ops(ctx, op):
    P = ref.P
    r = rng(31)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 380, (1 << 380) - 1, (1 << 28) - 1, 1 << 28, 0xFFFFFFF << 28, (1 << 64) - 1, (1 << 364) - 1]
    vals = edge + [r.randrange(P) for _ in range(3000)]
    a = b"".join(b48(v) for v in vals)
    b = b"".join(b48(vals[(7 * i + 3) % len(vals)]) for i in range(len(vals)))
    mul, add = (lambda x, y: cpu.fq_op("mul", x, y)), (lambda x, y: cpu.fq_op("add", x, y))
    want = {"mul_asm": lambda: mul(a, b), "sqr_asm": lambda: mul(a, a), "mul2_asm": lambda: add(mul(a, b), mul(b, a)),
            "mul_asm_lazy": lambda: mul(add(a, b), add(a, a)), "sqr_asm_lazy": lambda: mul(add(a, b), add(a, b))}
    assert ctx.fq_op(op, a, b) == (want[op]() if op in want else cpu.fq_op(op, a, b))


def test_point_ops_and_special_cases(ctx):
    r = rng(32)
    pts = ref.sample_points(33, 12)

    def jac(pt):
        if pt is None:
            return bytes(JB)
        z = r.randrange(1, ref.P)
        return b48(pt[0] * z * z % ref.P) + b48(pt[1] * z * z * z % ref.P) + b48(z)

    def aff(raw):
        return [cpu.to_affine64(raw[i:i + JB]) for i in range(0, len(raw), JB)]

    a = pts[:6] + [None, pts[3], pts[4], pts[5], None]
    b = pts[6:12] + [pts[2], None, pts[4], ref.neg(pts[5]), None]
    A, B = b"".join(jac(x) for x in a), b"".join(jac(x) for x in b)
    assert aff(ctx.g1_op("add", A, B)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, b)]
    assert aff(ctx.g1_op("double", A)) == [ref.affine_to_bytes64(ref.add(x, x)) for x in a]
    q = pts[1:7] + [pts[6], pts[3], ref.neg(pts[4]), pts[5], pts[7]]
    Q = ref.points_to_bytes(q)
    assert aff(ctx.g1_op("add_affine", A, Q)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert aff(ctx.g1_op("madd_w_pmp", A, Q)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert aff(ctx.g1_op("madd_w_mm", A, Q)) == [ref.affine_to_bytes64(ref.add(ref.add(x, ref.neg(y)), ref.neg(y))) for x, y in zip(a, q)]
    ks = [0, 1, 2, 3, 0xFFFF, 0x8000, 0xFFFFFFFF, 12345, 7, 1 << 31, 99]
    got = ctx.g1_mul_u32(A, ks)
    assert aff(got) == [ref.affine_to_bytes64(ref.mul(k, x)) for k, x in zip(ks, a)]


@pytest.mark.parametrize("n", [1, 3, 257, 4097, 70001])
def test_msm_matches_oracle_every_window_size_and_mode(ctx, n):
    points, sc = cpu.sample_points(34, n), bytearray(cpu.sample_scalars(35, n))
    edge = [0, 1, ref.R - 1, ref.R - 2, 0x8000, (1 << 254) - 1]
    for i, v in enumerate(edge[: min(n, len(edge))]):
        sc[32 * i:32 * i + 32] = b32(v)
    if n > 40:  # duplicates and negated duplicates of a point (the mixed addition's doubling / cancellation paths)
        pt = bytearray(points)
        pt[PB * 20:PB * 21] = pt[PB * 21:PB * 22]
        y = int.from_bytes(pt[PB * 22 + CB:PB * 23], "little")
        pt[PB * 23:PB * 24] = pt[PB * 22:PB * 22 + CB] + b48(ref.P - y)
        sc[32 * 20:32 * 21] = sc[32 * 21:32 * 22]
        sc[32 * 23:32 * 24] = sc[32 * 22:32 * 23]
        points = bytes(pt)
    sc = bytes(sc)
    want = cpu.to_affine64(cpu.cpu_msm(points, sc, 8))
    dev = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    ctx.set_bases(points, check_on_curve=True)
    for bits in (0, 12, 14, 16):
        ctx.set_window_bits(bits)
        assert ctx.msm(sc).to_affine_bytes() == want, (n, bits)
    ctx.set_window_bits(0)
    assert ctx.msm(dev).to_affine_bytes() == want
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc * 3, n)] == [want] * 3
    # window shards combine to the whole (the multi-GPU decomposition), fixed-base tables, the in-process multi-GPU ABI
    parts = [ctx.msm_windows(dev, 0, 6), ctx.msm_windows(dev, 6, 16)]
    assert parts[0].shape == (6, JB)
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0), curve="bls12_381").to_affine_bytes() == want
    if n <= 4097:
        ctx.set_bases(points, precompute=True)
        assert ctx.msm(sc).to_affine_bytes() == want
        ctx.set_bases(points, precompute="wide")          # 14 digits of 19 bits (top digit shifted by 10 on this curve), 8 virtual windows
        assert ctx.msm(sc).to_affine_bytes() == want and ctx.msm(dev).to_affine_bytes() == want
        assert ctx.wide_bits() == 16                      # (the policy's width for a small base set; beyond 2^16 points 19 bits here, not 17:)
        ctx.set_wide_bits(17)                             # 15 digits of 17 bits cannot hold this scalar field (1.8 x 2^254)
        try:
            with pytest.raises(m.MsmHipError) as e:
                ctx.set_bases(points, precompute="wide")
            assert e.value.code == -2
            for bits in (19, 20):
                ctx.set_wide_bits(bits)
                ctx.set_bases(points, precompute="wide")
                assert ctx.wide_bits() == bits and ctx.msm(sc).to_affine_bytes() == want
        finally:
            ctx.set_wide_bits(0)
        ctx.set_bases(points)
        mg = m.MultiGpuMsm([0, 0, 0], "host", curve="bls12_381")
        try:
            mg.set_bases(points)
            assert mg.msm(sc).to_affine_bytes() == want
            mg.launch_batch(sc * 2, n, 1)
            assert [g.to_affine_bytes() for g in mg.finish_batch(1, 2)] == [want] * 2
        finally:
            mg.close()


@pytest.mark.parametrize("n", [5, 3000, 20000])
def test_endomorphism_mode_with_points_of_order_r(ctx, n):
    points = subgroup_points(50, n)
    sc = bytearray(cpu.sample_scalars(51, n))
    lam = ref.glv_params()["lam"]
    for i, v in enumerate([0, 1, ref.R - 1, lam, ref.R - lam][: min(n, 5)]):
        sc[32 * i:32 * i + 32] = b32(v)
    sc = bytes(sc)
    want = cpu.to_affine64(cpu.cpu_msm(points, sc, 8))
    ctx.set_bases(points, check_on_curve=True, endomorphism=True)
    assert ctx.uses_endomorphism()
    for bits in (0, 12, 14, 16):
        ctx.set_window_bits(bits)
        assert ctx.msm(sc).to_affine_bytes() == want, bits
    ctx.set_window_bits(0)
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc * 3, n)] == [want] * 3
    # the half-length window shares of an 8-rank run, gathered in rank order
    dev = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    out = torch.zeros((8, JB), dtype=torch.uint8, device=dev.device)
    for rank in range(8):
        ctx.launch_half_windows_batch(dev, n, rank, rank + 1, rank % 3, out[rank:rank + 1])
        ctx.slot_sync(rank % 3)
    assert m.MsmContext.combine_windows(out, curve="bls12_381").to_affine_bytes() == want
    ctx.set_bases(points)
    assert ctx.msm(sc).to_affine_bytes() == want


def test_samplers_and_large_msm(ctx):
    n = 1 << 17
    pts, sc = ctx.sample_points(n, 36), ctx.sample_scalars(n, 37)
    assert pts.shape == (n, PB)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    assert pb[: PB * 300] == cpu.sample_points(36, 300) and sb[: 32 * 300] == cpu.sample_scalars(37, 300)
    assert cpu.points_on_curve(pb[: PB * 2000])
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, 8))
    ctx.set_bases(pts)
    assert ctx.msm(sc).to_affine_bytes() == want
    # skew: every scalar equal, and a witness-like vector
    s = (0x1234_5678_9ABC_DEF0_1357_9BDF_2468_ACE0_FEDC_BA98_7654_3210 * 0x10001) % ref.R
    eq = b32(s) * n
    assert ctx.msm(eq).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, eq, 8))
    gen = torch.Generator(device=pts.device)
    gen.manual_seed(3)
    sel = torch.rand(n, device=pts.device, generator=gen)
    wit = sc.clone()
    wit[sel < 0.7] = 0
    wit[(sel >= 0.4) & (sel < 0.7), 0] = 1
    assert ctx.msm(wit).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, wit.cpu().numpy().tobytes(), 8))


def test_2p20_points_bit_exact(ctx):
    # BASELINE config 2's size on this curve against the multithreaded oracle; the stage read-back of one window against the CPU stage models
    n = 1 << 20
    pts, sc = ctx.sample_points(n, 40), ctx.sample_scalars(n, 41)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, max(1, min(16, os.cpu_count() or 1))))
    ctx.set_bases(pts)
    assert ctx.msm(sc).to_affine_bytes() == want
    ctx.launch(sc, 0)
    ctx.launch(sc, 1)
    assert ctx.finish(0).to_affine_bytes() == want and ctx.finish(1).to_affine_bytes() == want
    ctx.set_bases(pts[:4].contiguous())


def test_stages_against_the_cpu_stage_models(ctx):
    # (≙ tests/smvp_shader.rs:292-334, tests/cuzk.rs: bucket sums and window sums of the device against the reference's CPU models)
    import numpy as np

    n = 40000
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
    for w in (0, 15):
        cp, vi = cpu.transpose(digits[w], 1 << 16)
        want = cpu.smvp_signed(cp, vi, points, 1 << 16)
        got = buckets[w].tobytes()
        assert [cpu.to_affine64(got[i:i + JB]) for i in range(0, len(got), JB)] == [cpu.to_affine64(want[i:i + JB]) for i in range(0, len(want), JB)]
        assert cpu.to_affine64(wsums[w].tobytes()) == cpu.to_affine64(cpu.bucket_reduction("running_sum", got))
    assert cpu.to_affine64(cpu.horner(wsums.tobytes(), 16)) == result.to_affine_bytes()
    assert result.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, scalars))
    # without debug the single-MSM launch leaves bit-plane sums: the read-back finishes them
    ctx.msm(scalars)
    assert cpu.to_affine64(cpu.horner(ctx.read_window_sums(16).tobytes(), 16)) == result.to_affine_bytes()
    assert isinstance(np.asarray(wsums), np.ndarray) and wsums.shape == (16, JB)


def test_input_errors(ctx):
    pts = cpu.sample_points(38, 4)
    with pytest.raises(m.MsmHipError):  # coordinate >= p
        ctx.set_bases(b48(ref.P) + pts[CB:])
    with pytest.raises(m.MsmHipError):  # not on the curve
        ctx.set_bases(b48(5) + b48(7) + pts[PB:], check_on_curve=True)
    ctx.set_bases(pts)
    with pytest.raises(m.MsmHipError):  # a scalar that overflows the 16-bit recode
        ctx.msm(b"\xff" * 32 + bytes(96))
    assert ctx.msm(bytes(128)).is_identity()
    out = m.lib().msm_hip_ctx_curve(ctx._h)
    assert out == 4
