"""The G2 models (oracle/bn254_g2_ref.py and its BLS12-381 instance) pinned: public parameters, field identities, and their two MSM
algorithms against each other and against the closed form over known multiples of the generator."""
import importlib

import pytest

from oracle import bls12_381_ref, bn254_ref


@pytest.fixture(params=["bn254_g2", "bls12_381_g2"], autouse=True)
def model(request):
    global g2
    g2 = importlib.import_module("oracle." + request.param + "_ref")
    return g2


def test_parameters_and_generator():
    g1 = bn254_ref if g2.FB == 32 else bls12_381_ref
    assert g2.P == g1.P and g2.R == g1.R  # the same base and scalar fields as the curve's G1 (BN254: src/cuzk/msm.rs:39, src/naive/utils/bigint.rs:85)
    assert g2.f2_mul((0, 1), (0, 1)) == (g2.P - 1, 0)  # u^2 = -1
    if g2.FB == 32:
        assert g2.f2_mul(g2.B, (9, 1)) == (3, 0)  # BN254: the twist's constant is 3 / (9 + u)
    else:
        assert g2.B == g2.f2_scale((1, 1), 4)  # BLS12-381: 4 (1 + u)
    assert g2.is_on_curve(g2.G)
    assert g2.mul(g2.R, g2.G) is g2.INF and g2.add(g2.mul(g2.R - 1, g2.G), g2.G) is g2.INF  # the generator has order r
    assert g2.mul(2, g2.G) == g2.add(g2.G, g2.G) and g2.is_on_curve(g2.mul(2, g2.G))
    a = (123456789, 987654321)
    assert g2.f2_mul(a, g2.f2_inv(a)) == (1, 0)


def test_jacobian_and_affine_formulas_agree():
    p, q = g2.mul(5, g2.G), g2.mul(7, g2.G)
    jp, jq = g2.j_from_affine(p), g2.j_from_affine(q)
    assert g2.j_to_affine(g2.j_add(jp, jq)) == g2.mul(12, g2.G)
    assert g2.j_to_affine(g2.j_add(jp, jp)) == g2.mul(10, g2.G)  # equal inputs take the doubling branch
    assert g2.j_to_affine(g2.j_add(jp, g2.j_from_affine(g2.neg(p)))) is g2.INF
    assert g2.j_to_affine(g2.j_double(g2.JINF)) is g2.INF and g2.j_to_affine(g2.j_add(g2.JINF, jq)) == q


@pytest.mark.parametrize("n,seed", [(1, 1), (3, 2), (33, 3)])
def test_the_two_msm_algorithms_agree(n, seed):
    pts = g2.sample_points(n, seed)
    assert all(g2.is_on_curve(p) for p in pts) and len(set(pts)) == n
    sc = [g2.sample_scalar(seed + 100, i) for i in range(n)]
    assert g2.msm_naive(pts, sc) == g2.msm_pippenger(pts, sc) == g2.msm_by_multipliers(g2.sample_multipliers(n, seed), sc)
    edge = [0, 1, g2.R - 1, 0x8000, 0x7fff, (1 << 253) + 0x80008000][:n]
    assert g2.msm_naive(pts[: len(edge)], edge) == g2.msm_pippenger(pts[: len(edge)], edge)


def test_wire_format_round_trip():
    pts = g2.sample_points(4, 9)
    b = g2.points_to_bytes(pts)
    fb = g2.FB
    assert len(b) == 4 * 4 * fb and g2.bytes_to_points(b) == pts
    assert b[:fb] == pts[0][0][0].to_bytes(fb, "little") and b[fb:2 * fb] == pts[0][0][1].to_bytes(fb, "little")  # c0 || c1
    j = g2.f2_to_bytes(pts[1][0]) + g2.f2_to_bytes(pts[1][1]) + g2.f2_to_bytes((1, 0))
    assert g2.jacobian_bytes_to_affine(j) == pts[1] and g2.jacobian_bytes_to_affine(bytes(6 * fb)) is g2.INF


def test_the_c_restatement_agrees_with_the_python_model():
    # oracle/bn254.c -DORACLE_G2 (the restatement of the reference's generic CPU MSM and of its stage models, over Fq2) against this model:
    # field ops, group ops, the synthetic points byte for byte, the MSM (serial and threaded), the cuZK stage models end to end
    cpu = importlib.import_module("oracle.cpu_" + g2.__name__.split(".")[-1].replace("_ref", ""))
    import random

    r = random.Random(5)
    P, cb = g2.P, g2.CB
    assert cpu.coord_bytes() == cb
    vals = [(0, 0), (1, 0), (0, 1), (P - 1, P - 1), (P - 1, 0)] + [(r.randrange(P), r.randrange(P)) for _ in range(200)]
    other = vals[3:] + vals[:3]
    A, B = b"".join(g2.f2_to_bytes(v) for v in vals), b"".join(g2.f2_to_bytes(v) for v in other)
    model = {"add": g2.f2_add, "sub": g2.f2_sub, "mul": g2.f2_mul, "sqr": lambda a, b: g2.f2_sqr(a), "neg": lambda a, b: g2.f2_neg(a)}
    for name, f in model.items():
        assert cpu.fq_op(name, A, B) == b"".join(g2.f2_to_bytes(f(a, b)) for a, b in zip(vals, other)), name
    nz = [v for v in vals if v != (0, 0)]
    assert cpu.fq_op("inv", b"".join(g2.f2_to_bytes(v) for v in nz), None) == b"".join(g2.f2_to_bytes(g2.f2_inv(v)) for v in nz)
    n = 40
    pts = g2.sample_points(n, 21)
    pb = cpu.sample_points(21, n)
    assert pb == g2.points_to_bytes(pts) and cpu.points_on_curve(pb) and not cpu.points_on_curve(pb[:cb] + pb[:cb] + pb[2 * cb:])
    assert cpu.sample_points(21, 10, first=30) == pb[30 * 2 * cb:]  # the sequence can be entered anywhere
    sb = cpu.sample_scalars(22, n)
    sc = g2.bytes_to_scalars(sb)
    want = g2.affine_to_bytes(g2.msm_pippenger(pts, sc))
    assert cpu.to_affine64(cpu.cpu_msm(pb, sb)) == want == cpu.to_affine64(cpu.cpu_msm(pb, sb, 3)) == cpu.to_affine64(cpu.msm_cuzk_model(pb, sb))
    jac = cpu.g1_scalar_mul(pb, sb)
    assert [g2.jacobian_bytes_to_affine(jac[3 * cb * i:3 * cb * (i + 1)]) for i in range(n)] == [g2.mul(k, p_) for k, p_ in zip(sc, pts)]
    assert cpu.to_affine64(cpu.horner(jac[: 16 * 3 * cb], 16)) == g2.affine_to_bytes(g2.msm_naive([g2.mul(k, p_) for k, p_ in zip(sc, pts)][:16], [1 << (16 * w) for w in range(16)]))


def test_both_models_reproduce_the_committed_golden_vectors():
    # tests/golden/msm_vectors_g2.json (written by tests/golden/make_golden_g2.py): the C restatement on every case, the Python model on the
    # small ones and, for the seeded cases, through the closed form
    from tests.util import case_inputs_g2, golden_cases_g2

    name = g2.__name__.split(".")[-1].replace("_ref", "")
    cpu = importlib.import_module("oracle.cpu_" + name)
    cases = [c for c in golden_cases_g2() if c["curve"] == name]
    assert len(cases) == 14
    for case in cases:
        pb, sb = case_inputs_g2(case)
        want = bytes.fromhex(case["expected_affine"])
        assert cpu.to_affine64(cpu.cpu_msm(pb, sb, 4)) == want, case["name"]
        sc = g2.bytes_to_scalars(sb)
        if len(sc) <= 20:
            assert g2.affine_to_bytes(g2.msm_naive(g2.bytes_to_points(pb), sc)) == want, case["name"]
        if case["kind"] == "seeded":
            assert g2.affine_to_bytes(g2.msm_by_multipliers(g2.sample_multipliers(case["n"], case["point_seed"]), sc)) == want, case["name"]
