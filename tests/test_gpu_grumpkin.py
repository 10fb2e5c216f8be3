"""A second curve behind the same ABI and the same kernels (the reference is hard-wired to BN254's Fq,
src/cuzk/msm.rs:37-43, and lists other curves as future work): Grumpkin, BN254's cycle partner (y^2 = x^3 - 17 over BN254's
scalar field, scalars modulo BN254's base field).  Checked against the Grumpkin builds of both oracles
(oracle/cpu_grumpkin.py = oracle/bn254.c with -DORACLE_GRUMPKIN; oracle/grumpkin_ref.py = the big-integer model)."""
import numpy as np
import pytest
import torch

import msm_webgpu_amd as m
from oracle import cpu_grumpkin as cpu
from oracle import grumpkin_ref as ref
from tests.util import rng

pytestmark = pytest.mark.gpu
P, R = ref.P, ref.R  # base-field and scalar-field moduli of Grumpkin


@pytest.fixture(scope="module")
def gctx(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = m.MsmContext(0, curve="grumpkin")
    yield c
    c.close()


def b32(x):
    return int(x).to_bytes(32, "little")


def jac(pt, r):
    if pt is None:
        return bytes(96)
    z = r.randrange(1, P)
    return b32(pt[0] * z * z % P) + b32(pt[1] * z * z * z % P) + b32(z)


def aff64(xyz):
    return [cpu.to_affine64(xyz[i:i + 96]) for i in range(0, len(xyz), 96)]


def test_oracles_agree_and_know_the_generator():
    # public known answer: Grumpkin's generator is (1, sqrt(-16)) with this y; (r - 1) G = -G
    assert ref.G == (1, 17631683881184975370165255887551781615748388533673675138860)
    assert ref.mul(ref.R - 1, ref.G) == ref.neg(ref.G)
    pts, sc = ref.sample_points(3, 9), ref.sample_scalars(4, 9)
    assert cpu.sample_points(3, 9) == ref.points_to_bytes(pts) and cpu.sample_scalars(4, 9) == ref.scalars_to_bytes(sc)
    want = ref.affine_to_bytes64(ref.msm_naive(pts, sc))
    assert cpu.to_affine64(cpu.cpu_msm(ref.points_to_bytes(pts), ref.scalars_to_bytes(sc))) == want
    assert cpu.to_affine64(cpu.msm_cuzk_model(ref.points_to_bytes(pts), ref.scalars_to_bytes(sc), 16)) == want


@pytest.mark.parametrize("op", ["add", "sub", "mul", "sqr", "neg", "mul_asm", "sqr_asm", "mul2_asm", "mul_asm_lazy", "sqr_asm_lazy"])
def test_field_ops(gctx, op):
    r = rng(21)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 253, (1 << 253) - 1, (1 << 29) - 1, 1 << 29, 0x1FFFFFFF << 29, (1 << 64) - 1]
    vals = edge + [r.randrange(P) for _ in range(3000)]
    a = b"".join(b32(v) for v in vals)
    b = b"".join(b32(vals[(7 * i + 3) % len(vals)]) for i in range(len(vals)))
    mul, add = (lambda x, y: cpu.fq_op("mul", x, y)), (lambda x, y: cpu.fq_op("add", x, y))
    want = {"mul_asm": lambda: mul(a, b), "sqr_asm": lambda: mul(a, a), "mul2_asm": lambda: add(mul(a, b), mul(b, a)),
            "mul_asm_lazy": lambda: mul(add(a, b), add(a, a)), "sqr_asm_lazy": lambda: mul(add(a, b), add(a, b))}
    assert gctx.fq_op(op, a, b) == (want[op]() if op in want else cpu.fq_op(op, a, b))


def test_point_ops(gctx):
    r = rng(22)
    pts = ref.sample_points(31, 40)
    a = pts[:16] + [None, pts[3], pts[4], pts[5], None]
    b = pts[16:32] + [pts[2], None, pts[4], ref.neg(pts[5]), None]
    A, B = b"".join(jac(x, r) for x in a), b"".join(jac(x, r) for x in b)
    assert aff64(gctx.g1_op("add", A, B)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, b)]
    assert aff64(gctx.g1_op("double", A)) == [ref.affine_to_bytes64(ref.add(x, x)) for x in a]
    q = pts[20:36] + [pts[6], pts[3], ref.neg(pts[4]), pts[5], pts[7]]
    Q = ref.points_to_bytes(q)
    assert aff64(gctx.g1_op("add_affine", A, Q)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert aff64(gctx.g1_op("madd_w_pmp", A, Q)) == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert aff64(gctx.g1_op("madd_w_mm", A, Q)) == [ref.affine_to_bytes64(ref.add(ref.add(x, ref.neg(y)), ref.neg(y))) for x, y in zip(a, q)]


def test_explicit_cases(gctx):
    G, pts = ref.G, ref.sample_points(41, 8)
    cases = [([G], [2]), ([G], [R - 1]), ([G], [0]), ([pts[0], pts[0]], [7, R - 7]), (pts[:4], [0, 0, 12345, 0]),
             ([pts[1]] * 5, [3] * 5), ([pts[2], ref.neg(pts[2]), pts[3]], [9, 9, 1]), (pts[:2], [int("8000" * 15, 16), 0x8000]),
             (pts[:8], [R - 1, R - 2, 1, 2, (1 << 253) - 1, 0xFFFF, 0x7FFF, 1 << 250])]
    for points, scalars in cases:
        gctx.set_bases(ref.points_to_bytes(points), check_on_curve=True)
        got = gctx.msm(ref.scalars_to_bytes(scalars))
        assert got.to_affine_bytes() == ref.affine_to_bytes64(ref.msm_naive(points, scalars)), scalars
    assert gctx.msm(ref.scalars_to_bytes([5])).to_affine() == ref.mul(5, pts[0])  # G1.to_affine uses Grumpkin's modulus
    with pytest.raises(m.MsmHipError) as e:  # BN254's generator (1, 2) is not on Grumpkin
        gctx.set_bases(b32(1) + b32(2), check_on_curve=True)
    assert e.value.code == -5
    with pytest.raises(m.MsmHipError) as e:  # a coordinate >= Grumpkin's base modulus (but below BN254's)
        gctx.set_bases(b32(P) + b32(2))
    assert e.value.code == -4


@pytest.mark.parametrize("n", [1, 65, 1000, 4097, 50000, 1 << 17])
def test_msm_matches_oracle(gctx, n):
    points, scalars = cpu.sample_points(500 + n, n), cpu.sample_scalars(501 + n, n)
    want = cpu.to_affine64(cpu.cpu_msm(points, scalars, 8))
    gctx.set_bases(points)
    assert gctx.msm(scalars).to_affine_bytes() == want                                              # host scalars
    assert gctx.msm(torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()).to_affine_bytes() == want


def test_device_samplers_and_every_mode(gctx):
    n = 6000
    pts, sc = gctx.sample_points(n, 77), gctx.sample_scalars(n, 78)  # Tonelli-Shanks on the device (r = 1 mod 4)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    assert pb == cpu.sample_points(77, n) and sb == cpu.sample_scalars(78, n)
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, 8))
    gctx.set_bases(pts, check_on_curve=True)
    for bits in (12, 14, 16):
        gctx.set_window_bits(bits)
        assert gctx.msm(sc).to_affine_bytes() == want
    gctx.set_window_bits(0)
    parts = [gctx.msm_windows(sc, 0, 5), gctx.msm_windows(sc, 5, 16)]                                # window shards + host combine
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0), curve="grumpkin").to_affine_bytes() == want
    wsums = gctx.msm_windows(sc, 0, 16).cpu().numpy().tobytes()                                       # per-window sums vs the stage model
    digits = cpu.decompose_scalars_signed(sb, 16, 16)
    for w in (0, 15):
        cp, vi = cpu.transpose(digits[w], 1 << 16)
        buckets = cpu.smvp_signed(cp, vi, pb, 1 << 16)
        assert cpu.to_affine64(wsums[96 * w:96 * w + 96]) == cpu.to_affine64(cpu.bucket_reduction("running_sum", buckets))
    gctx.set_bases(pts, precompute="wide")                                                            # wide fixed-base tables (19-bit digits)
    assert gctx.msm(sc).to_affine_bytes() == want
    gctx.set_bases(pts, precompute=True)                                                              # fixed-base tables
    assert gctx.msm(sc).to_affine_bytes() == want
    batch = torch.cat([sc, sc.flip(0).contiguous()], dim=0).contiguous()
    got = gctx.msm_batch(batch, n)
    assert got[0].to_affine_bytes() == want
    assert got[1].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sc.flip(0).contiguous().cpu().numpy().tobytes(), 8))


def test_both_curves_side_by_side(gctx, ctx):
    # a BN254 context and a Grumpkin context in one process give their own curve's results on the same bytes where valid
    n = 300
    from oracle import cpu as cpu_bn

    pb_bn, sb = cpu_bn.sample_points(90, n), cpu_bn.sample_scalars(91, n)  # scalars < r < p: valid for both curves
    pb_gr = cpu.sample_points(90, n)
    ctx.set_bases(pb_bn)
    gctx.set_bases(pb_gr)
    assert ctx.msm(sb).to_affine_bytes() == cpu_bn.to_affine64(cpu_bn.cpu_msm(pb_bn, sb))
    assert gctx.msm(sb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb_gr, sb))


def test_montgomery_word_formats(gctx):
    # MSM_HIP_BASES_MONT256 / MSM_HIP_SCALARS_MONT256 with Grumpkin's moduli (base field r, scalar field p of BN254)
    n = 500
    points, scalars = cpu.sample_points(600, n), cpu.sample_scalars(601, n)
    want = cpu.to_affine64(cpu.cpu_msm(points, scalars))
    pm = b"".join(((x << 256) % P).to_bytes(32, "little") + ((y << 256) % P).to_bytes(32, "little") for x, y in ref.bytes_to_points(points))
    sm = b"".join(((v << 256) % R).to_bytes(32, "little") for v in ref.bytes_to_scalars(scalars))
    gctx.set_bases(pm, check_on_curve=True, mont256=True)
    gctx.set_scalar_format(True)
    try:
        assert gctx.msm(sm).to_affine_bytes() == want
    finally:
        gctx.set_scalar_format(False)
    gctx.set_bases(points)
    assert gctx.msm(scalars).to_affine_bytes() == want


def test_multi_gpu_abi_on_the_second_curve(built):
    # msm_hip_mgpu_create_curve: the in-process multi-GPU shape (window shares + gather + ONE combine; batches as whole MSMs) on
    # Grumpkin -- several contexts on the one GPU, as tests/test_gpu_mgpu.py does for BN254
    n = 20000
    points, scalars = cpu.sample_points(950, n), cpu.sample_scalars(951, n)
    want = cpu.to_affine64(cpu.cpu_msm(points, scalars, 8))
    for ids, gather in (([0, 0, 0], "host"), ([0], "rccl")):
        mg = m.MultiGpuMsm(ids, gather, curve="grumpkin")
        try:
            mg.set_bases(points, check_on_curve=True)
            assert mg.msm(scalars).to_affine_bytes() == want
            got = mg.msm_batch(scalars[: 32 * 6000] , 2000)
            assert len(got) == 3
            assert got[1].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[: 64 * 2000], scalars[32 * 2000: 32 * 4000]))
        finally:
            mg.close()
