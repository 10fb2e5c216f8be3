"""CPU build (g++ -DFQ_CHECK) of the exact arithmetic headers the HIP kernels inline -- fq29.h (9 x 29-bit lazy limbs)
and g1.h (XYZZ formulas) -- checked against the oracle, with every limb/value bound asserted.  Host logic only."""
import ctypes as C
import os
import subprocess

import pytest

from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import P, b32, jacobian_bytes, rng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fq29") / "fq29_harness.so")
    # MSM_TEST_SANITIZE=1: the same headers under UBSan (signed overflow, shifts, alignment ...), aborting on the first report
    san = ["-fsanitize=undefined", "-fno-sanitize-recover=all"] if os.environ.get("MSM_TEST_SANITIZE") == "1" else []
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFQ_CHECK", "-fPIC", "-shared"] + san + ["-I", os.path.join(ROOT, "msm-webgpu_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "fq29_harness.cpp"), "-o", so])
    return C.CDLL(so)


def test_field_ops(H):
    r = rng(1)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 253, (1 << 253) - 1, 1 << 232, (1 << 232) - 1, (1 << 29) - 1, 1 << 29,
            0x1FFFFFFF << 29]
    vals = edge + [r.randrange(P) for _ in range(3000)]
    n = len(vals)
    A = b"".join(b32(x) for x in vals)
    B = b"".join(b32(vals[(i * 7 + 3) % n]) for i in range(n))
    for op, name in enumerate(["add", "sub", "mul", "sqr", "neg"]):
        out = C.create_string_buffer(32 * n)
        H.h_fq_op(op, A, B, out, n)
        assert out.raw == cpu.fq_op(name, A, B), name


def test_pack_unpack_roundtrip(H):
    r = rng(2)
    raw = b"".join(r.randrange(1 << 256).to_bytes(32, "little") for _ in range(200)) + b"\xff" * 32 + bytes(32)
    out = C.create_string_buffer(len(raw))
    H.h_fq_roundtrip(raw, out, len(raw) // 32)
    assert out.raw == raw


def test_mixed_add_chain_and_special_cases(H):
    r = rng(3)
    pts = ref.sample_points(5, 40)
    out = C.create_string_buffer(96)
    chain = [pts[0], pts[0], pts[1], ref.neg(pts[1]), pts[2], pts[2], pts[2], pts[3]] + pts[4:30]
    want = None
    for q in chain:
        want = ref.add(want, q)
    H.h_g1_madd_chain(bytes(96), ref.points_to_bytes(chain), len(chain), out)
    assert cpu.to_affine64(out.raw) == ref.affine_to_bytes64(want)
    H.h_g1_madd_chain(jacobian_bytes(ref.neg(pts[7]), r), ref.points_to_bytes([pts[7]]), 1, out)
    assert cpu.to_affine64(out.raw) == bytes(64)
    H.h_g1_madd_chain(jacobian_bytes(ref.neg(pts[7]), r), ref.points_to_bytes([pts[7], pts[8]]), 2, out)
    assert cpu.to_affine64(out.raw) == ref.affine_to_bytes64(pts[8])
    H.h_g1_madd_chain(jacobian_bytes(pts[9], r), ref.points_to_bytes([pts[9]]), 1, out)
    assert cpu.to_affine64(out.raw) == ref.affine_to_bytes64(ref.add(pts[9], pts[9]))


def test_long_accumulation_keeps_bounds(H):
    lp = cpu.sample_points(77, 5000)
    out = C.create_string_buffer(96)
    H.h_g1_madd_chain(bytes(96), lp, 5000, out)
    assert cpu.to_affine64(out.raw) == cpu.to_affine64(cpu.cpu_msm(lp, b32(1) * 5000))


def test_full_add_double_scalar(H):
    r = rng(4)
    pts = ref.sample_points(6, 24)
    a = pts[:10] + [None, pts[3], pts[4], pts[5], None]
    b = pts[10:20] + [pts[2], None, pts[4], ref.neg(pts[5]), None]
    A = b"".join(jacobian_bytes(x, r) for x in a)
    B = b"".join(jacobian_bytes(x, r) for x in b)
    o = C.create_string_buffer(96 * len(a))
    H.h_g1_op(0, A, B, o, len(a))
    for i in range(len(a)):
        assert cpu.to_affine64(o.raw[96 * i:96 * i + 96]) == ref.affine_to_bytes64(ref.add(a[i], b[i])), i
    H.h_g1_op(1, A, None, o, len(a))
    for i in range(len(a)):
        assert cpu.to_affine64(o.raw[96 * i:96 * i + 96]) == ref.affine_to_bytes64(ref.add(a[i], a[i])), i
    out = C.create_string_buffer(96)
    for k in [0, 1, 2, 3, 255, 256, 32767, 0xFFFFFFFF, 123456789]:
        H.h_g1_mul_u32(jacobian_bytes(pts[11], r), k, out)
        assert cpu.to_affine64(out.raw) == ref.affine_to_bytes64(ref.mul(k, pts[11])), k


def test_running_sum_feedback(H):
    r = rng(5)
    bk = ref.sample_points(8, 40)
    out = C.create_string_buffer(96)
    H.h_g1_running_sum(b"".join(jacobian_bytes(x, r) for x in bk), len(bk), out)
    want = None
    for i, q in enumerate(bk):
        want = ref.add(want, ref.mul(len(bk) - i, q))
    assert cpu.to_affine64(out.raw) == ref.affine_to_bytes64(want)


def test_generated_headers_are_in_sync():
    # bn254_constants.h and fq29_asm.h are generated; the committed files must be what the generators emit
    import sys

    for script, args, header in (("gen_constants.py", ["bls12_381"], "bls12_381_constants.h"), ("gen_constants.py", ["bn254_g2"], "bn254_g2_constants.h"), ("gen_fq29_asm.py", ["14", "28", "254"], "fq28x14_asm.h"),
                                 ("gen_constants.py", [], "bn254_constants.h"), ("gen_constants.py", ["grumpkin"], "grumpkin_constants.h"),
                                 ("gen_constants.py", ["pallas"], "pallas_constants.h"), ("gen_constants.py", ["vesta"], "vesta_constants.h"),
                                 ("gen_fq29_asm.py", [], "fq29_asm.h")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, capture_output=True, text=True, check=True).stdout
        assert out == open(os.path.join(ROOT, "msm-webgpu_amd", "csrc", header)).read(), header


def test_signed_state_mixed_addition_chain(H):
    # g1_madd_w (the SMVP's form: the accumulator keeps +-Y, the sign is applied at the flush) against affine big-integer
    # arithmetic: random signs, the doubling and cancellation cases in both sign states, long chains (bounds asserted: -DFQ_CHECK)
    r = rng(6)
    pts = ref.sample_points(9, 40)
    out = C.create_string_buffer(96)

    def run(acc, chain, negs):
        H.h_g1_madd_w_chain(jacobian_bytes(acc, r), ref.points_to_bytes(chain), bytes(negs), len(chain), out)
        want = acc
        for q, ng in zip(chain, negs):
            want = ref.add(want, ref.neg(q) if ng else q)
        assert cpu.to_affine64(out.raw) == ref.affine_to_bytes64(want), (negs,)

    for trial in range(30):
        k = r.randrange(1, 12)
        run(None if trial % 3 == 0 else pts[trial], [pts[r.randrange(40)] for _ in range(k)], [r.randrange(2) for _ in range(k)])
    a, b = pts[0], pts[1]
    run(None, [a, a], [0, 0])            # P + P from the fresh (unsigned) state: doubling
    run(None, [a, b, ref.add(a, b)], [0, 0, 0])   # doubling met in the signed state (after two steps W = -Y ... +Y)
    run(None, [a, b, ref.add(a, b)], [0, 0, 1])   # cancellation in that state
    run(None, [a, a], [0, 1])            # P - P
    run(None, [a, a, b], [1, 0, 0])      # -P + P = identity, then a fresh start
    run(b, [a, ref.add(a, b)], [0, 1])   # (b + a) - (a + b)
    run(b, [a, ref.add(a, b)], [0, 0])   # (b + a) + (a + b): doubling with W = -Y
    lp = cpu.sample_points(78, 4000)
    negs = bytes(r.randrange(2) for _ in range(4000))
    H.h_g1_madd_w_chain(bytes(96), lp, negs, 4000, out)
    sc = b"".join(b32(ref.R - 1 if ng else 1) for ng in negs)
    assert cpu.to_affine64(out.raw) == cpu.to_affine64(cpu.cpu_msm(lp, sc))


def _check_bounds(Hx, m, seed):
    # the formulas of g1.h with their inputs at the edge of the promised value bounds (X < 9p, Y < 5p, ZZ, ZZZ < 2p): larger
    # representatives of the same residues must give the same points, and no multiplication may leave [0, 2p) (asserted in the harness)
    r = rng(seed)
    pts = m.sample_points(seed, 12)
    cb = getattr(m, "CB", 32)  # bytes per coordinate (48: BLS12-381)
    out = C.create_string_buffer(3 * cb)
    b32 = lambda v: int(v).to_bytes(cb, "little")

    def jac(pt):
        if pt is None:
            return bytes(3 * cb)
        z = r.randrange(1, m.P)
        return b32(pt[0] * z * z % m.P) + b32(pt[1] * z * z * z % m.P) + b32(z)

    def aff(raw):
        x, y, z = (int.from_bytes(raw[k:k + cb], "little") for k in (0, cb, 2 * cb))
        if z == 0:
            return None
        zi = pow(z, -1, m.P)
        return (x * zi * zi % m.P, y * zi * zi * zi % m.P)

    for trial in range(40):
        a, b = pts[trial % 12], pts[(trial * 5 + 1) % 12]
        for kx, ky, kz in ((8, 4, 1), (8, 0, 0), (0, 4, 1), (7, 3, 0), (0, 0, 0)):
            neg = trial & 1
            # op 0 with the sign state set: the stored Y stands for -Y, so the accumulator handed over is -a
            for wneg in (0, 1):
                Hx.h_g1_at_the_bounds(0, jac(m.neg(a) if wneg else a), m.points_to_bytes([b]), neg, wneg, kx, ky, kz, out)
                assert aff(out.raw) == m.add(a, m.neg(b) if neg else b), (trial, kx, ky, kz, wneg)
            Hx.h_g1_at_the_bounds(1, jac(a), m.points_to_bytes([b]), 0, 0, kx, ky, kz, out)
            assert aff(out.raw) == m.add(a, b)
            Hx.h_g1_at_the_bounds(2, jac(a), jac(b), 0, 0, kx, ky, kz, out)
            assert aff(out.raw) == m.add(a, b)
            Hx.h_g1_at_the_bounds(3, jac(a), None, 0, 0, kx, ky, kz, out)
            assert aff(out.raw) == m.add(a, a)
        # the special cases at the bounds: P + P and P - P through the mixed additions
        Hx.h_g1_at_the_bounds(0, jac(a), m.points_to_bytes([a]), 0, 0, 8, 4, 1, out)
        assert aff(out.raw) == m.add(a, a)
        Hx.h_g1_at_the_bounds(0, jac(a), m.points_to_bytes([a]), 1, 0, 8, 4, 1, out)
        assert aff(out.raw) is None


def test_formulas_at_the_edge_of_their_value_bounds(H):
    _check_bounds(H, ref, 12)


def _check_glv(Hx, m, seed):
    # csrc/glv.h against the oracle's independently derived model: identical halves (same rounding rule), k = k1 + k2 lambda,
    # and the magnitude bound the 8-window recode needs; edge scalars, non-canonical scalars up to 2^256 - 1
    r = rng(seed)
    q = m.glv_params()
    ks = [0, 1, 2, m.R - 1, m.R, m.R + 1, (1 << 256) - 1, 1 << 255, q["lam"], m.R - q["lam"], (1 << 128) - 1, 1 << 127]
    ks += [r.randrange(m.R) for _ in range(4000)] + [r.randrange(1 << 256) for _ in range(500)]
    # scalars next to a rounding boundary of c1 / c2 (k ~ (j + 1/2) r / |b|)
    for b in (abs(q["v1"][1]), abs(q["v2"][1])):
        for _ in range(200):
            j = r.randrange(b)
            k0 = ((2 * j + 1) * m.R) // (2 * b)
            ks += [k for k in (k0 - 1, k0, k0 + 1) if 0 <= k < 1 << 256]
    out = C.create_string_buffer(32 * len(ks))
    Hx.h_glv_split.restype = C.c_size_t
    assert Hx.h_glv_split(b"".join(k.to_bytes(32, "little") for k in ks), out, len(ks)) == 0
    for i, k in enumerate(ks):
        got = []
        for j in range(2):
            v = int.from_bytes(out.raw[32 * i + 16 * j:32 * i + 16 * j + 16], "little")
            mag = v & ((1 << 127) - 1)
            assert mag < (1 << 127) - (1 << 112)
            got.append(-mag if v >> 127 else mag)
        assert tuple(got) == m.glv_split(k), hex(k)
        assert (got[0] + got[1] * q["lam"] - k) % m.R == 0
    xs = [r.randrange(m.P) for _ in range(50)]
    cb = getattr(m, "CB", 32)
    o = C.create_string_buffer(cb * len(xs))
    Hx.h_fq_mul_beta(b"".join(x.to_bytes(cb, "little") for x in xs), o, len(xs))
    assert o.raw == b"".join((q["beta"] * x % m.P).to_bytes(cb, "little") for x in xs)
    pt = m.mul(0x1234567, m.G)  # a point of order r (phi = lambda only there: BLS12-381's sampled curve points carry cofactor components)
    assert m.mul(q["lam"], pt) == m.endo(pt)


def test_endomorphism_scalar_split(H):
    _check_glv(H, ref, 10)


def test_host_arithmetic_instantiated_for_grumpkin(tmp_path_factory):
    # the same headers compiled for the second curve (csrc/curve_select.h), bounds asserted, against the Grumpkin oracle
    from oracle import cpu_grumpkin as cg
    from oracle import grumpkin_ref as gr

    so = str(tmp_path_factory.mktemp("fq29g") / "fq29_harness_grumpkin.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFQ_CHECK", "-fPIC", "-shared", "-DMSM_FIELD_NS=grumpkin", "-DMSM_KERNEL_NS=msmk_grumpkin",
                           '-DMSM_CURVE_CONSTANTS="grumpkin_constants.h"', "-DHARNESS_FIELD_NS=grumpkin", "-I", os.path.join(ROOT, "msm-webgpu_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "fq29_harness.cpp"), "-o", so])
    Hg = C.CDLL(so)
    r = rng(8)
    vals = [0, 1, gr.P - 1, gr.P - 2, (1 << 253) - 1, 1 << 232] + [r.randrange(gr.P) for _ in range(2000)]
    n = len(vals)
    A = b"".join(b32(x) for x in vals)
    B = b"".join(b32(vals[(i * 7 + 3) % n]) for i in range(n))
    for op, name in enumerate(["add", "sub", "mul", "sqr", "neg"]):
        out = C.create_string_buffer(32 * n)
        Hg.h_fq_op(op, A, B, out, n)
        assert out.raw == cg.fq_op(name, A, B), name
    lp = cg.sample_points(79, 3000)
    negs = bytes(r.randrange(2) for _ in range(3000))
    out = C.create_string_buffer(96)
    Hg.h_g1_madd_w_chain(bytes(96), lp, negs, 3000, out)
    sc = b"".join(b32(gr.R - 1 if ng else 1) for ng in negs)
    assert cg.to_affine64(out.raw) == cg.to_affine64(cg.cpu_msm(lp, sc))
    _check_glv(Hg, gr, 11)
    _check_bounds(Hg, gr, 13)


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_host_arithmetic_instantiated_for_the_pasta_curves(tmp_path_factory, curve):
    # 255-bit moduli: 2^261 / p = 127 instead of 169.  The same headers, every limb bound and every Montgomery result (< 2p) asserted:
    # field ops, a long signed-state accumulation, the formulas at the edge of their value bounds, the endomorphism split
    import importlib

    cx = importlib.import_module("oracle.cpu_" + curve)
    rf = importlib.import_module("oracle." + curve + "_ref")
    so = str(tmp_path_factory.mktemp("fq29" + curve) / ("fq29_harness_%s.so" % curve))
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFQ_CHECK", "-fPIC", "-shared", "-DMSM_FIELD_NS=" + curve, "-DMSM_KERNEL_NS=msmk_" + curve,
                           '-DMSM_CURVE_CONSTANTS="%s_constants.h"' % curve, "-DHARNESS_FIELD_NS=" + curve, "-I", os.path.join(ROOT, "msm-webgpu_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "fq29_harness.cpp"), "-o", so])
    Hc = C.CDLL(so)
    r = rng(20)
    vals = [0, 1, rf.P - 1, rf.P - 2, (1 << 254) - 1, 1 << 254, 1 << 232] + [r.randrange(rf.P) for _ in range(2000)]
    n = len(vals)
    A = b"".join(b32(x) for x in vals)
    B = b"".join(b32(vals[(i * 7 + 3) % n]) for i in range(n))
    for op, name in enumerate(["add", "sub", "mul", "sqr", "neg"]):
        out = C.create_string_buffer(32 * n)
        Hc.h_fq_op(op, A, B, out, n)
        assert out.raw == cx.fq_op(name, A, B), name
    lp = cx.sample_points(81, 3000)
    negs = bytes(r.randrange(2) for _ in range(3000))
    out = C.create_string_buffer(96)
    Hc.h_g1_madd_w_chain(bytes(96), lp, negs, 3000, out)
    sc = b"".join(b32(rf.R - 1 if ng else 1) for ng in negs)
    assert cx.to_affine64(out.raw) == cx.to_affine64(cx.cpu_msm(lp, sc))
    _check_glv(Hc, rf, 21)
    _check_bounds(Hc, rf, 22)


def test_host_arithmetic_instantiated_for_bls12_381(tmp_path_factory):
    # the same headers with ANOTHER LIMB LAYOUT: 14 limbs of 28 bits, 12 packed words, 48-byte coordinates (csrc/bls12_381_constants.h).
    # Every limb bound and every Montgomery result (< 2p) asserted: field ops at the edges of the 381-bit field, pack / unpack, a long
    # signed-state accumulation against the BLS12-381 oracle, the group formulas at the edge of their value bounds, the endomorphism split
    from oracle import bls12_381_ref as rf
    from oracle import cpu_bls12_381 as cx

    so = str(tmp_path_factory.mktemp("fq28bls") / "fq_harness_bls12_381.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFQ_CHECK", "-fPIC", "-shared", "-DMSM_FIELD_NS=bls12_381", "-DMSM_KERNEL_NS=msmk_bls12_381",
                           '-DMSM_CURVE_CONSTANTS="bls12_381_constants.h"', "-DHARNESS_FIELD_NS=bls12_381", "-I", os.path.join(ROOT, "msm-webgpu_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "fq29_harness.cpp"), "-o", so])
    Hc = C.CDLL(so)
    r = rng(30)
    b48 = lambda v: int(v).to_bytes(48, "little")
    vals = [0, 1, rf.P - 1, rf.P - 2, (1 << 380) - 1, 1 << 380, 1 << 364, (1 << 28) - 1, 1 << 28, (1 << 381) - 1 - (1 << 381) % 1] + [r.randrange(rf.P) for _ in range(2000)]
    vals = [v % rf.P for v in vals]
    n = len(vals)
    A = b"".join(b48(x) for x in vals)
    B = b"".join(b48(vals[(i * 7 + 3) % n]) for i in range(n))
    for op, name in enumerate(["add", "sub", "mul", "sqr", "neg"]):
        out = C.create_string_buffer(48 * n)
        Hc.h_fq_op(op, A, B, out, n)
        assert out.raw == cx.fq_op(name, A, B), name
    raw = b"".join(r.randrange(1 << 384).to_bytes(48, "little") for _ in range(500))
    out = C.create_string_buffer(len(raw))
    Hc.h_fq_roundtrip(raw, out, 500)
    assert out.raw == raw
    lp = cx.sample_points(82, 2000)
    negs = bytes(r.randrange(2) for _ in range(2000))
    out = C.create_string_buffer(144)
    Hc.h_g1_madd_w_chain(bytes(144), lp, negs, 2000, out)
    # (the sampler's points lie on the curve, not necessarily in the order-r subgroup -- the cofactor is not 1 -- so "-P" is not (r - 1) P
    #  here: the expected sum is formed with explicit negations in the big-integer model)
    acc = None
    for pt, ng in zip(rf.bytes_to_points(lp), negs):
        acc = rf.add(acc, rf.neg(pt) if ng else pt)
    assert cx.to_affine64(out.raw) == rf.affine_to_bytes64(acc)
    _check_glv(Hc, rf, 31)
    _check_bounds(Hc, rf, 32)


@pytest.mark.parametrize("curve", ["bn254_g2", "bls12_381_g2"])
def test_host_arithmetic_instantiated_for_g2(tmp_path_factory, curve):
    # a G2 unit: csrc/fq2.h (Fq2 on the curve's prime field -- 9 x 29 or 14 x 28-bit limbs --, every sum reduced) under the same group formulas
    # (g1.h), built as csrc/curve_<curve>.hip builds it, every limb bound and every Montgomery result asserted (-DFQ_CHECK), against the G2 model
    import importlib

    g2 = importlib.import_module("oracle." + curve + "_ref")
    JREC = 3 * g2.CB
    so = str(tmp_path_factory.mktemp("fq2" + curve) / "fq2_harness.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFQ_CHECK", "-fPIC", "-shared"] + (["-DHARNESS_G2_BLS12_381"] if curve == "bls12_381_g2" else []) +
                          ["-I", os.path.join(ROOT, "msm-webgpu_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "host_harness"),
                           os.path.join(ROOT, "tests", "host_harness", "fq2_harness.cpp"), "-o", so])
    Hc = C.CDLL(so)
    r = rng(40)
    P = g2.P
    edge = [0, 1, P - 1, P - 2, (P - 1) // 2, 1 << (P.bit_length() - 1), (1 << 29) - 1, 1 << 28]
    vals = [(a, b) for a in edge for b in edge] + [(r.randrange(P), r.randrange(P)) for _ in range(1500)]
    n = len(vals)
    A = b"".join(g2.f2_to_bytes(v) for v in vals)
    other = [vals[(i * 7 + 3) % n] for i in range(n)]
    B = b"".join(g2.f2_to_bytes(v) for v in other)
    model = {"add": g2.f2_add, "sub": g2.f2_sub, "mul": g2.f2_mul, "sqr": lambda a, b: g2.f2_sqr(a), "neg": lambda a, b: g2.f2_neg(a)}
    for op, name in enumerate(["add", "sub", "mul", "sqr", "neg"]):
        out = C.create_string_buffer(g2.CB * n)
        Hc.h_fq_op(op, A, B, out, n)
        assert out.raw == b"".join(g2.f2_to_bytes(model[name](a, b)) for a, b in zip(vals, other)), name
    raw = b"".join(r.randrange(1 << (8 * g2.FB)).to_bytes(g2.FB, "little") for _ in range(2 * 300))
    out = C.create_string_buffer(len(raw))
    Hc.h_fq_roundtrip(raw, out, 300)
    assert out.raw == raw
    # a long signed-state accumulation (the SMVP's inner operation), with a doubling (the point meets itself) and a cancellation inside
    pts = g2.sample_points(300, 5)
    chain = pts[:100] + [pts[7]] + pts[100:200] + [g2.neg(pts[150])] + pts[200:]
    negs = bytes(r.randrange(2) for _ in range(len(chain)))
    out = C.create_string_buffer(JREC)
    Hc.h_g1_madd_w_chain(bytes(JREC), g2.points_to_bytes(chain), negs, len(chain), out)
    acc = None
    for pt, ng in zip(chain, negs):
        acc = g2.add(acc, g2.neg(pt) if ng else pt)
    assert g2.jacobian_bytes_to_affine(out.raw) == acc
    # P + P, P - P and -P + P through the signed form from an empty accumulator; the plain mixed addition likewise
    for seq, sg, want in (([pts[0], pts[0]], b"\0\0", g2.add(pts[0], pts[0])), ([pts[0], pts[0]], b"\0\1", None), ([pts[0], pts[0]], b"\1\0", None),
                          ([pts[0], pts[0], pts[1]], b"\1\1\0", g2.add(g2.neg(g2.add(pts[0], pts[0])), pts[1]))):
        Hc.h_g1_madd_w_chain(bytes(JREC), g2.points_to_bytes(seq), sg, len(seq), out)
        assert g2.jacobian_bytes_to_affine(out.raw) == want
    Hc.h_g1_madd_chain(bytes(JREC), g2.points_to_bytes(pts[:50] + [pts[3]]), 51, out)
    want = None
    for pt in pts[:50] + [pts[3]]:
        want = g2.add(want, pt)
    assert g2.jacobian_bytes_to_affine(out.raw) == want
    # full additions / doublings of Jacobian inputs with z != 1, scalar multiples, running sums
    def jac(pt, k):
        z = (r.randrange(1, P), r.randrange(P)) if k else (1, 0)
        if pt is None:
            return bytes(JREC)
        z2 = g2.f2_sqr(z)
        return g2.f2_to_bytes(g2.f2_mul(pt[0], z2)) + g2.f2_to_bytes(g2.f2_mul(pt[1], g2.f2_mul(z2, z))) + g2.f2_to_bytes(z)

    pa = [pts[i] for i in range(40)] + [None, pts[0], pts[1]]
    pb = [pts[i + 40] for i in range(40)] + [pts[2], pts[0], g2.neg(pts[1])]
    JA, JB = b"".join(jac(p_, 1) for p_ in pa), b"".join(jac(p_, i % 2) for i, p_ in enumerate(pb))
    out = C.create_string_buffer(JREC * len(pa))
    Hc.h_g1_op(0, JA, JB, out, len(pa))
    assert [g2.jacobian_bytes_to_affine(out.raw[JREC * i:JREC * i + JREC]) for i in range(len(pa))] == [g2.add(a, b) for a, b in zip(pa, pb)]
    Hc.h_g1_op(1, JA, None, out, len(pa))
    assert [g2.jacobian_bytes_to_affine(out.raw[JREC * i:JREC * i + JREC]) for i in range(len(pa))] == [g2.add(a, a) for a in pa]
    o1 = C.create_string_buffer(JREC)
    for k in (0, 1, 2, 0x7fff, 0x8000, 0xffffffff):
        Hc.h_g1_mul_u32(jac(pts[9], 1), k, o1)
        assert g2.jacobian_bytes_to_affine(o1.raw) == g2.mul(k, pts[9]), k
    Hc.h_g1_running_sum(b"".join(jac(p_, 1) for p_ in pts[:64]), 64, o1)
    want = None
    for i, p_ in enumerate(pts[:64]):
        want = g2.add(want, g2.mul(64 - i, p_))
    assert g2.jacobian_bytes_to_affine(o1.raw) == want
