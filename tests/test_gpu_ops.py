"""Device field / point operations vs the oracle (≙ the reference's tests/field.rs:110-166 and tests/point.rs:125-186,
which compare single-op shaders with halo2curves on random operands).  Bit-exact; seeded; plus edge operands."""
import numpy as np
import pytest

from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import P, affine64_list, b32, jacobian_bytes, rng

pytestmark = pytest.mark.gpu

EDGE = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 253, (1 << 253) - 1, (1 << 232) - 1, 1 << 232,
        (1 << 29) - 1, 1 << 29, 0x1FFFFFFF << 29, (1 << 64) - 1, 1 << 64]


def _operands(n, seed):
    r = rng(seed)
    vals = EDGE + [r.randrange(P) for _ in range(n)]
    other = [vals[(7 * i + 3) % len(vals)] for i in range(len(vals))]
    return b"".join(b32(v) for v in vals), b"".join(b32(v) for v in other)


@pytest.mark.parametrize("op", ["add", "sub", "mul", "sqr", "neg"])
def test_field_ops_match_oracle(ctx, op):
    a, b = _operands(5000, 11)
    assert ctx.fq_op(op, a, b) == cpu.fq_op(op, a, b)


def test_montgomery_round_trip(ctx):
    # x * 1 == x : into and out of the device Montgomery form (≙ tests/field.rs:140-152)
    a, _ = _operands(2000, 12)
    one = b32(1) * (len(a) // 32)
    assert ctx.fq_op("mul", a, one) == a


def _points(n, seed):
    return ref.bytes_to_points(cpu.sample_points(seed, n))


def test_point_add_double_and_special_cases(ctx):
    r = rng(13)
    pts = _points(64, 31)
    a = pts[:32] + [None, pts[3], pts[4], pts[5], None]
    b = pts[32:64] + [pts[2], None, pts[4], ref.neg(pts[5]), None]  # inf+Q, P+inf, P+P, P+(-P), inf+inf
    A = b"".join(jacobian_bytes(x, r) for x in a)
    B = b"".join(jacobian_bytes(x, r) for x in b)
    assert affine64_list(ctx.g1_op("add", A, B)) == affine64_list(cpu.g1_op("add", A, B))
    assert affine64_list(ctx.g1_op("double", A)) == affine64_list(cpu.g1_op("double", A))
    for i, (x, y) in enumerate(zip(a, b)):
        assert affine64_list(ctx.g1_op("add", A, B))[i] == ref.affine_to_bytes64(ref.add(x, y)), i


def test_point_mixed_add(ctx):
    r = rng(14)
    pts = _points(40, 32)
    a = pts[:20] + [None, pts[7], ref.neg(pts[8])]
    q = pts[20:40] + [pts[6], pts[7], pts[8]]  # inf + Q, P + P (doubling), -P + P (identity)
    A = b"".join(jacobian_bytes(x, r) for x in a)
    Q = ref.points_to_bytes(q)
    got = affine64_list(ctx.g1_op("add_affine", A, Q))
    want = [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]
    assert got == want


def test_double_and_add_small_scalars(ctx):
    # ≙ tests/point.rs:170-185 (double_and_add with a u32 scalar)
    r = rng(15)
    pts = _points(12, 33)
    ks = [0, 1, 2, 3, 255, 256, 32767, 32768, 65535, 0xFFFFFFFF, 123456789, 1 << 31]
    A = b"".join(jacobian_bytes(x, r) for x in pts)
    got = affine64_list(ctx.g1_mul_u32(A, np.array(ks, dtype=np.uint32)))
    want = [ref.affine_to_bytes64(ref.mul(k, p)) for k, p in zip(ks, pts)]
    assert got == want


@pytest.mark.parametrize("op", ["mul_asm", "sqr_asm", "mul2_asm", "mul_asm_lazy", "sqr_asm_lazy"])
def test_assembly_multipliers_match_oracle(ctx, op):
    # fq_mul_asm / fq_sqr_asm / fq_mul2_asm (csrc/fq29_asm.h) called directly, on the edge operand set and on lazy limbs
    a, b = _operands(20000, 16)
    mul, add = (lambda x, y: cpu.fq_op("mul", x, y)), (lambda x, y: cpu.fq_op("add", x, y))
    want = {"mul_asm": lambda: mul(a, b), "sqr_asm": lambda: mul(a, a), "mul2_asm": lambda: add(mul(a, b), mul(b, a)),
            "mul_asm_lazy": lambda: mul(add(a, b), add(a, a)), "sqr_asm_lazy": lambda: mul(add(a, b), add(a, b))}[op]()
    assert ctx.fq_op(op, a, b) == want


def test_signed_state_mixed_addition(ctx):
    # g1_madd_w, the SMVP's inner operation (W = +-Y state, lazily negated point, sign applied at the flush), every sign state
    r = rng(17)
    pts = _points(60, 34)
    a = pts[:25] + [None, pts[7], ref.neg(pts[8]), pts[9], None]
    q = pts[25:50] + [pts[6], pts[7], pts[8], ref.neg(pts[9]), pts[10]]
    A = b"".join(jacobian_bytes(x, r) for x in a)
    Q = ref.points_to_bytes(q)
    got = affine64_list(ctx.g1_op("madd_w_pmp", A, Q))
    assert got == [ref.affine_to_bytes64(ref.add(x, y)) for x, y in zip(a, q)]                      # a + q - q + q
    got = affine64_list(ctx.g1_op("madd_w_mm", A, Q))
    assert got == [ref.affine_to_bytes64(ref.add(ref.add(x, ref.neg(y)), ref.neg(y))) for x, y in zip(a, q)]  # a - q - q
