"""A check against code that none of this repository's models share anything with: sympy's `EllipticCurve` (affine chord-and-tangent
formulas over Python integers, written by the sympy authors, part of this image).  The reference holds no MSM or group-law vectors
(src/lib.rs:21-37 draws its inputs from `thread_rng`), so the oracle's group law is otherwise pinned only by the repository's own
two models and a few public constants (tests/test_oracle.py); here the C oracle of every G1 curve -- and, on the GPU, the HIP path itself -- must
give sympy's sum for the same (scalar, point) lists, including the scalars 0, 1, r - 1, r (wire format: canonical, so r is rejected or reduced by
the caller -- not used), equal points (doubling inside the sum) and opposite points (an identity inside the sum).
Small n: sympy needs ~0.2 s per 254-bit scalar multiplication."""
import importlib

import pytest
from sympy.ntheory.elliptic_curve import EllipticCurve

from tests.util import rng

# curve constant b of y^2 = x^3 + b (public parameters of the curves; a = 0 everywhere)
CURVES = {"bn254": ("oracle.cpu", 3), "grumpkin": ("oracle.cpu_grumpkin", -17), "pallas": ("oracle.cpu_pallas", 5), "vesta": ("oracle.cpu_vesta", 5),
          "bls12_381": ("oracle.cpu_bls12_381", 4)}


def _case(curve, n, seed):
    """(oracle module, sympy curve, points as ints, scalars as ints, wire bytes of both): sampled points with two planted relations"""
    cpu = importlib.import_module(CURVES[curve][0])
    c = cpu.constants()
    p, r, cb = c["p"], c["r"], cpu.coord_bytes()
    E = EllipticCurve(0, CURVES[curve][1] % p, modulus=p)
    raw = cpu.sample_points(seed, n)
    pts = [(int.from_bytes(raw[2 * cb * i:2 * cb * i + cb], "little"), int.from_bytes(raw[2 * cb * i + cb:2 * cb * (i + 1)], "little")) for i in range(n)]
    if n >= 6:
        pts[3] = pts[1]                      # the same point twice: a doubling somewhere in the sum
        pts[5] = (pts[4][0], p - pts[4][1])  # a point and its negative
    rr = rng(seed + 1)
    sc = [rr.randrange(r) for _ in range(n)]
    if n >= 6:
        sc[0], sc[2] = 0, r - 1
        sc[3] = sc[1]                        # s P + s P
        sc[5] = sc[4]                        # s P + s (-P) = identity
    if n >= 8:
        sc[6], sc[7] = 1, (1 << 128) + 5
    pb = b"".join(x.to_bytes(cb, "little") + y.to_bytes(cb, "little") for x, y in pts)
    sb = b"".join(s.to_bytes(32, "little") for s in sc)
    return cpu, E, pts, sc, pb, sb, cb


def _sympy_msm(E, pts, sc):
    acc = None
    for (x, y), s in zip(pts, sc):
        P = E(x, y)                          # (sympy checks nothing here; the sum below only makes sense on the curve, which the assert covers)
        assert (y * y - x * x * x - int(E._a6)) % int(E.modulus) == 0
        t = s * P
        acc = t if acc is None else acc + t
    return acc


_EXPECTED = {}  # (curve, n, seed) -> sympy's sum: ~0.2 s per scalar multiplication, the same case serves every base mode


def _case_with_expected(curve, n, seed):
    case = _case(curve, n, seed)
    key = (curve, n, seed)
    if key not in _EXPECTED:
        _EXPECTED[key] = _expect(_sympy_msm(case[1], case[2], case[3]))
    return case + (_EXPECTED[key],)


def _affine_ints(cpu, xyz, cb):
    a = cpu.to_affine64(xyz)
    return int.from_bytes(a[:cb], "little"), int.from_bytes(a[cb:], "little")


def _expect(acc):
    """sympy's point as (x, y), (0, 0) for the identity (the wire form of to_affine on z = 0)"""
    if acc is None or int(acc.z) == 0:
        return (0, 0)
    return int(acc.x), int(acc.y)


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_c_oracle_equals_sympy(curve):
    cpu, E, pts, sc, pb, sb, cb = _case(curve, 8, 4100)
    want = _expect(_sympy_msm(E, pts, sc))
    assert want != (0, 0)
    assert _affine_ints(cpu, cpu.cpu_msm(pb, sb), cb) == want
    assert _affine_ints(cpu, cpu.msm_cuzk_model(pb, sb, 16), cb) == want   # the stage models (decompose -> transpose -> SMVP -> reduce -> Horner) too
    # single scalar multiplications and an identity result
    assert _affine_ints(cpu, cpu.g1_scalar_mul(pb[:2 * cb], sb[32:64]), cb) == _expect(sc[1] * E(*pts[0]))
    assert _affine_ints(cpu, cpu.cpu_msm(pb[8 * cb:12 * cb], sb[4 * 32:6 * 32]), cb) == (0, 0)   # s P + s (-P)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", sorted(CURVES))
@pytest.mark.parametrize("mode", ["default", "plain", "tables_wide"])
def test_hip_path_equals_sympy(built, curve, mode):
    import torch

    import msm_webgpu_amd as m

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cpu, E, pts, sc, pb, sb, cb, want = _case_with_expected(curve, 12, 4200)
    ctx = m.MsmContext(0, curve=curve)
    try:
        if mode == "default":
            ctx.set_bases(pb, endomorphism=None)    # flags = 0: the endomorphism mode where the curve has prime order
        elif mode == "plain":
            ctx.set_bases(pb, endomorphism=False)
        else:
            ctx.set_bases(pb, precompute="wide")
        got = ctx.msm(sb).to_affine()
        assert (got or (0, 0)) == want
        # the planted pair alone: s P + s (-P) is the identity on the device too
        ctx.set_bases(pb[8 * cb:12 * cb], endomorphism=None if mode == "default" else False, precompute="wide" if mode == "tables_wide" else False)
        assert ctx.msm(sb[4 * 32:6 * 32]).to_affine() is None
    finally:
        ctx.close()
