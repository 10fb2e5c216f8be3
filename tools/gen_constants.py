#!/usr/bin/env python3
"""Generate msm-webgpu_amd/csrc/bn254_constants.h: BN254 Fq constants for the 9 x 29-bit limb, R = 2^261
Montgomery representation used by the HIP kernels, and the 4 x 64-bit, R = 2^256 constants used by the
host finalisation.  Values are derived from p alone (decimal in /root/reference/src/cuzk/msm.rs:39;
the reference derives its own 13-bit-limb constants at run time in src/cuzk/utils.rs:283-373).
Run: python tools/gen_constants.py > msm-webgpu_amd/csrc/bn254_constants.h
     python tools/gen_constants.py grumpkin > msm-webgpu_amd/csrc/grumpkin_constants.h
     python tools/gen_constants.py pallas   > msm-webgpu_amd/csrc/pallas_constants.h      (likewise vesta, bls12_381)
     python tools/gen_constants.py bn254_g2 > msm-webgpu_amd/csrc/bn254_g2_constants.h   (a G2 unit's Fq2-level constants; likewise bls12_381_g2)
(Grumpkin, BN254's cycle partner: base field = BN254's scalar field, scalar field = BN254's base field, y^2 = x^3 - 17; the two
moduli agree in their top 128 bits, so 2^261 / modulus = 169 and every lazy bound of fq29.h / g1.h holds for both.
Pallas / Vesta, the Pasta cycle: 255-bit moduli, 2^261 / modulus = 127 -- the largest operand product of the group formulas,
P^2 with P < 11.02 p in the mixed addition, is 121.5 p^2: DESIGN.md section 4.10.)
"""
import sys

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
RMOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
CURVE, CURVE_B, CURVE_TEXT = "bn254", 3, "BN254 base field Fq"
GEN = (1, 2)
if len(sys.argv) > 1 and sys.argv[1] == "grumpkin":
    P, RMOD = RMOD, P
    CURVE, CURVE_B, CURVE_TEXT = "grumpkin", -17, "Grumpkin base field (= BN254 scalar field Fr)"
    GEN = (1, 17631683881184975370165255887551781615748388533673675138860)  # (1, sqrt(-16)), the smaller root
if len(sys.argv) > 1 and sys.argv[1] in ("pallas", "vesta"):
    PALLAS_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
    VESTA_P = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    P, RMOD = (PALLAS_P, VESTA_P) if sys.argv[1] == "pallas" else (VESTA_P, PALLAS_P)
    CURVE, CURVE_B = sys.argv[1], 5
    CURVE_TEXT = "Pallas base field Fp" if CURVE == "pallas" else "Vesta base field (= Pallas scalar field Fq)"
    GEN = (P - 1, 2)
W, L, WORDS = 29, 9, 8  # limb bits, limbs, packed 32-bit words per field element


def cube_root_of_unity(mod):
    g = 2
    while pow(g, (mod - 1) // 3, mod) == 1:
        g += 1
    return pow(g, (mod - 1) // 3, mod)


def glv_lattice(rmod, lam, shift=320):
    """The scalar split's constants (csrc/glv.h) for a cube root of unity `lam` modulo `rmod`: a short basis (a1, b1), (a2, b2) of
    {(x, y): x + y lam = 0 mod r} from the extended Euclidean sequence r_i = s_i r + t_i lam around sqrt(r) (Guide to Elliptic Curve
    Cryptography, algorithm 3.74), and from it  c1 = round(k b2 / r), c2 = round(-k b1 / r)  taken as sign * ((k * G + 2^(shift - 1)) >> shift)
    with G = round(2^shift |b| / r), and  (k1, k2) = (k, 0) - c1 V1 - c2 V2  modulo 2^160 in two's complement:
    k1 = k + m1 N11 + m2 N12, k2 = m1 N21 + m2 N22.  -> dict(V1, V2, G1, G2, N11, N12, N21, N22, ok)"""
    seq = [(rmod, 0), (lam, 1)]
    while seq[-1][0]:
        q = seq[-2][0] // seq[-1][0]
        seq.append((seq[-2][0] - q * seq[-1][0], seq[-2][1] - q * seq[-1][1]))
    l = max(i for i, (rem, _) in enumerate(seq) if rem * rem >= rmod)
    v1 = (seq[l + 1][0], -seq[l + 1][1])
    cands = [(seq[l][0], -seq[l][1]), (seq[l + 2][0], -seq[l + 2][1])]
    v2 = min(cands, key=lambda v: v[0] * v[0] + v[1] * v[1])
    if v1[0] * v2[1] - v2[0] * v1[1] < 0:
        v2 = (-v2[0], -v2[1])
    assert v1[0] * v2[1] - v2[0] * v1[1] == rmod
    for v in (v1, v2):
        assert (v[0] + v[1] * lam) % rmod == 0
    s1, s2 = (1 if v2[1] >= 0 else -1), (1 if -v1[1] >= 0 else -1)
    g1 = ((abs(v2[1]) << shift) + rmod // 2) // rmod
    g2 = ((abs(v1[1]) << shift) + rmod // 2) // rmod
    m160 = (1 << 160) - 1
    # |k1| <= (|a1| + |a2|) / 2 + (rounding slack), |k2| likewise; the recode of a 128-bit half needs < 2^127 - 2^112
    half_bound = max(abs(v1[0]) + abs(v2[0]), abs(v1[1]) + abs(v2[1])) * 513 // 1024 + 2
    ok = half_bound < (1 << 127) - (1 << 112) and g1 < 1 << 224 and g2 < 1 << 224
    return {"V1": v1, "V2": v2, "G1": g1, "G2": g2, "N11": (-s1 * v1[0]) & m160, "N12": (-s2 * v2[0]) & m160, "N21": (-s1 * v1[1]) & m160,
            "N22": (-s2 * v2[1]) & m160, "ok": ok}


def emit_g2(which):
    """msm-webgpu_amd/csrc/<curve>_g2_constants.h: the constants of a G2 unit at the level of its coordinate field Fq2 = Fq[u] / (u^2 + 1)
    (csrc/fq2.h; the prime field underneath is the G1 unit's <curve>_constants.h, instantiated in a namespace of its own).
    bn254: the twist y^2 = x^3 + 3 / (9 + u);  bls12_381: the twist y^2 = x^3 + 4 (1 + u).  Scalars modulo the same r as the curve's G1.
    Endomorphism mode (round 4): the twist has j-invariant 0 like the curve itself, so phi(x, y) = (beta x, y) with beta a cube root of unity of
    the PRIME field is an endomorphism of the twist too, and on G2 (order r) it is the multiplication by a cube root of unity lambda mod r --
    the same split k = k1 + k2 lambda, |k1|, |k2| < 2^127, and the same device code as G1 (csrc/glv.h), with the beta that belongs to lambda ON
    G2 (found here by checking lambda G = (beta x_G, y_G) on the subgroup's generator).  (The Frobenius-based psi is the other candidate; its
    eigenvalue t - 1 is 127 bits on BN254 but 64 bits on BLS12-381, where a balanced two-dimensional split needs psi^2 = (omega x, -y) -- the
    same map as phi up to sign.)  No device sampler."""
    if which == "bn254":
        p, rmod, w, l, nw = P, RMOD, 29, 9, 8
        n9 = pow(9 * 9 + 1, -1, p)  # 1 / (9 + u) = (9 - u) / 82
        b0, b1 = 3 * 9 * n9 % p, (-3 * n9) % p
        assert ((b0 * 9 - b1) % p, (b0 + 9 * b1) % p) == (3, 0)
        text = "BN254 G2: the twist y^2 = x^3 + 3 / (9 + u)"
        # the standard generator of G2 (EIP-197 / arkworks bn254; oracle/bn254_g2_ref.py pins it: on the twist, of order r)
        gen = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
               (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
    else:
        p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
        rmod = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
        w, l, nw = 28, 14, 12
        b0, b1 = 4, 4
        text = "BLS12-381 G2: the twist y^2 = x^3 + 4 (1 + u)"
        gen = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
               (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))
    assert p % 4 == 3  # u^2 = -1 is irreducible
    # affine arithmetic on the twist over Fq2 (elements (c0, c1)), for the one check below
    f2mul = lambda a, b_: ((a[0] * b_[0] - a[1] * b_[1]) % p, (a[0] * b_[1] + a[1] * b_[0]) % p)
    f2sub = lambda a, b_: ((a[0] - b_[0]) % p, (a[1] - b_[1]) % p)

    def f2inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * n % p, -a[1] * n % p)

    def g2add(a, b_):
        if a is None:
            return b_
        if b_ is None:
            return a
        if a[0] == b_[0]:
            if f2sub(a[1], b_[1]) != (0, 0):
                return None
            x2 = f2mul(a[0], a[0])
            lam_ = f2mul((3 * x2[0] % p, 3 * x2[1] % p), f2inv((2 * a[1][0] % p, 2 * a[1][1] % p)))
        else:
            lam_ = f2mul(f2sub(b_[1], a[1]), f2inv(f2sub(b_[0], a[0])))
        x3 = f2sub(f2sub(f2mul(lam_, lam_), a[0]), b_[0])
        return (x3, f2sub(f2mul(lam_, f2sub(a[0], x3)), a[1]))

    def g2mul(k, pt):
        acc = None
        while k:
            if k & 1:
                acc = g2add(acc, pt)
            pt = g2add(pt, pt)
            k >>= 1
        return acc

    assert f2sub(f2mul(gen[1], gen[1]), f2mul(f2mul(gen[0], gen[0]), gen[0])) == (b0 % p, b1 % p) and g2mul(rmod, gen) is None
    # lambda: the SAME cube root of unity mod r as the curve's G1 unit uses (so that the split constants are the G1 unit's, bit for bit);
    # beta: the cube root of unity mod p that belongs to it on G2
    beta1 = cube_root_of_unity(p)
    lam = cube_root_of_unity(rmod)
    g1gen = {"bn254": (1, 2)}.get(which) or (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                                             0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)

    def g1mul(k, pt):  # (x, y) over the prime field, y^2 = x^3 + b: only lambda's choice needs it
        def add(a, b_):
            if a is None:
                return b_
            if b_ is None:
                return a
            if a[0] == b_[0]:
                if (a[1] + b_[1]) % p == 0:
                    return None
                l_ = 3 * a[0] * a[0] * pow(2 * a[1], -1, p) % p
            else:
                l_ = (b_[1] - a[1]) * pow(b_[0] - a[0], -1, p) % p
            x = (l_ * l_ - a[0] - b_[0]) % p
            return (x, (l_ * (a[0] - x) - a[1]) % p)
        acc = None
        while k:
            if k & 1:
                acc = add(acc, pt)
            pt = add(pt, pt)
            k >>= 1
        return acc

    if g1mul(lam, g1gen) != (beta1 * g1gen[0] % p, g1gen[1]):
        lam = lam * lam % rmod
    assert g1mul(lam, g1gen) == (beta1 * g1gen[0] % p, g1gen[1])  # (the G1 unit's choice: tools/gen_constants.py without _g2)
    lg = g2mul(lam, gen)
    beta = next(bb for bb in (beta1, beta1 * beta1 % p) if lg == ((bb * gen[0][0] % p, bb * gen[0][1] % p), gen[1]))
    glv = glv_lattice(rmod, lam)
    assert glv["ok"]
    m = (1 << w) - 1
    rr = 1 << (w * l)
    hl = nw // 2
    lim = lambda x, n=l: [(x >> (w * i)) & m for i in range(n)]
    words = lambda x, bits, n: [(x >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]
    a = lambda name, vals, ty="uint32_t", fmt="0x%08xu": "static constexpr %s %s[%d] = {%s};" % (ty, name, len(vals), ", ".join(fmt % v for v in vals))
    print("// GENERATED by tools/gen_constants.py %s_g2 -- do not edit." % which)
    print("// %s over Fq2 = Fq[u] / (u^2 + 1), p = %d" % (text, p))
    print("// (the G2 unit's coordinate-field level: csrc/fq2.h builds Fq2 on the prime-field unit instantiated beside it)")
    print("#pragma once\n#include <cstdint>\nnamespace MSM_FIELD_NS {")
    print("constexpr int CURVE_B_SMALL = 0;  // (the twist's constant is not a small integer: FQ_B29)")
    print("constexpr int FQ_HEADROOM = %d;  // of the prime field: floor(2^%d / p) (capped)" % (min(rr // p, 1 << 20), w * l))
    print("constexpr int FQ_EXT = 2;  // degree of the coordinate field over the prime field")
    print("constexpr int FQ_LIMBS = %d;  // c0's %d limbs, then c1's\nconstexpr int FQ_LIMB_BITS = %d;\nconstexpr int FQ_WORDS = %d;  // packed 32-bit words per element (wire / storage form): c0 || c1\nconstexpr uint32_t FQ_MASK = 0x%xu;" % (2 * l, l, w, 2 * nw, m))
    print(a("FQ_P29", lim(p)) + "  // the prime field's modulus")
    print(a("FQ2_2P29", lim(2 * p)) + "  // 2p, exact limbs: the reduced additions of fq2.h")
    print(a("FQ_ONE29", lim(rr % p) + [0] * l) + "  // 1 = (R mod p, 0)")
    print(a("FQ_B29", lim(b0 * rr % p) + lim(b1 * rr % p)) + "  // the twist's constant, Montgomery form")
    print("// endomorphism phi(x, y) = (beta x, y) = lambda (x, y) on G2, beta in the prime field; k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^127: csrc/glv.h")
    print("// lambda = %d (the G1 unit's); beta = %d (%s the G1 unit's)" % (lam, beta, "=" if beta == beta1 else "the square of"))
    print(a("FQ_BETA29", lim(beta * rr % p) + [0] * l) + "  // (beta, 0), Montgomery form")
    print("// the standard generator of G2 (order r), Montgomery form: the device point sampler's base point (P_i = (a + i b) G, oracle/bn254_g2_ref.py: sample_points)")
    print(a("FQ_GEN_X29", lim(gen[0][0] * rr % p) + lim(gen[0][1] * rr % p)))
    print(a("FQ_GEN_Y29", lim(gen[1][0] * rr % p) + lim(gen[1][1] * rr % p)))
    print("// p as %d x 32-bit words: each component of a coordinate is compared with it" % nw)
    print(a("FQ_P32", words(p, 32, nw)))
    print("// host finalisation, prime-field level: %d x 64-bit limbs, R = 2^%d" % (hl, 64 * hl))
    f64 = "0x%016xull"
    rh = 1 << (64 * hl)
    print(a("FQ_P64", words(p, 64, hl), "uint64_t", f64))
    print("constexpr uint64_t FQ_N0_64 = 0x%016xull;  // -p^-1 mod 2^64" % ((-pow(p, -1, 1 << 64)) % (1 << 64)))
    print(a("FQ_ONE64", words(rh % p, 64, hl), "uint64_t", f64) + "  // 2^%d mod p" % (64 * hl))
    print(a("FQ_R2_64", words(rh * rh % p, 64, hl), "uint64_t", f64) + "  // 2^%d mod p" % (128 * hl))
    print(a("FR_R32", words(rmod, 32, 8)) + "  // scalar field modulus r")
    print(a("FR_R29", lim(rmod)))
    print("static constexpr uint32_t FR_N0_29 = 0x%08xu;" % ((-pow(rmod, -1, 1 << w)) % (1 << w)))
    print(a("FQ_PM2_32", words(p - 2, 32, nw)) + "  // p - 2 : the prime field's inversion exponent (the norm's inverse in an Fq2 inversion)")
    print("constexpr int FQ_BITS = %d;" % p.bit_length())
    print("constexpr bool GLV_SUPPORTED = true;  // MSM_HIP_BASES_ENDOMORPHISM available (bases of order r: the mode is an explicit choice on a curve with a cofactor)")
    print("constexpr int GLV_SHIFT = 320;")
    for name, n in (("G1", 7), ("G2", 7), ("N11", 5), ("N12", 5), ("N21", 5), ("N22", 5)):
        print(a("GLV_%s_32" % name, words(glv[name], 32, n)))
    print("}  // namespace MSM_FIELD_NS")


if len(sys.argv) > 1 and sys.argv[1] in ("bn254_g2", "bls12_381_g2"):
    emit_g2(sys.argv[1][:-3])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "bls12_381":
    # BLS12-381 G1: y^2 = x^3 + 4 over the 381-bit p, scalars modulo the 255-bit r (reference README.md "other curves").
    # 14 limbs of 28 bits (392 bits, R = 2^392): with 29-bit limbs the 28 partial products of a column (lazy squares: 14 x 2^60) overflow 64 bits.
    P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    RMOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    CURVE, CURVE_B, CURVE_TEXT = "bls12_381", 4, "BLS12-381 base field Fp"
    GEN = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
           0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
    W, L, WORDS = 28, 14, 12
assert (GEN[1] * GEN[1] - GEN[0] ** 3 - CURVE_B) % P == 0
RBITS = W * L  # device Montgomery radix R = 2^RBITS
HEADROOM = min((1 << RBITS) // P, 1 << 20)
assert HEADROOM >= 127, "a Montgomery product needs value(a) value(b) <= (2^261 / p) p^2; the group formulas need 121.5 p^2 (g1.h)"
# a column of a product holds L partial products of two lazy limbs (< 2^(W+1) + 16) and L reduction terms: it must fit 64 bits
assert L * ((1 << (W + 1)) + 64) ** 2 + L * (1 << W) ** 2 < 1 << 64, "column overflow"
HL = WORDS // 2  # 64-bit limbs of the host arithmetic
M = (1 << W) - 1


def limbs29(x, n=L):
    out = [(x >> (W * i)) & M for i in range(n)]
    assert sum(v << (W * i) for i, v in enumerate(out)) == x, "does not fit"
    return out


def limbs(x, bits, n):
    return [(x >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def borrow_proof(k):
    """k*p as L limbs c_i with c_i >= 2^31 - 4 for i < L - 1 so that (a_i + c_i - b_i) never goes negative
    for b limbs up to 2^31 - 4, and never exceeds 2^32 for a limbs up to 2^30 + 2^29."""
    q = limbs29(k * P)
    c = list(q)
    up = (1 << 31) >> W  # 2^31 lent to a limb is `up` units of the next one
    c[0] = q[0] + (1 << 31)
    for i in range(1, L - 1):
        c[i] = q[i] + (1 << 31) - up
    c[L - 1] = q[L - 1] - up
    assert sum(v << (W * i) for i, v in enumerate(c)) == k * P
    assert all(0 <= v < (1 << 32) for v in c) and c[L - 1] > 0
    return c


def arr(name, vals, ty="uint32_t", fmt="0x%08xu"):
    return "static constexpr %s %s[%d] = {%s};" % (ty, name, len(vals), ", ".join(fmt % v for v in vals))


R261 = 1 << RBITS  # ("261": the 9 x 29-bit layout's radix; RBITS in general)
print("// GENERATED by tools/gen_constants.py -- do not edit.")
print("// %s: p = %d" % (CURVE_TEXT, P))
print("// (included inside the field namespace of the curve unit being instantiated: csrc/curve_select.h)")
print("#pragma once\n#include <cstdint>\nnamespace MSM_FIELD_NS {")
print("constexpr int CURVE_B_SMALL = %d;  // y^2 = x^3 + b" % CURVE_B)
print("constexpr int FQ_HEADROOM = %d;  // floor(2^%d / p) (capped): a Montgomery product of operands with value(a) value(b) <= FQ_HEADROOM p^2 is < 2p" % (HEADROOM, RBITS))
print("constexpr int FQ_EXT = 1;  // degree of the coordinate field over the prime field (2 in the G2 unit, tools/gen_constants.py bn254_g2)")
print("constexpr int FQ_LIMBS = %d;\nconstexpr int FQ_LIMB_BITS = %d;\nconstexpr int FQ_WORDS = %d;  // packed 32-bit words per element (wire / storage form)\nconstexpr uint32_t FQ_MASK = 0x%xu;" % (L, W, WORDS, M))
print(arr("FQ_P29", limbs29(P)))
print("constexpr uint32_t FQ_N0_29 = 0x%xu;  // -p^-1 mod 2^%d" % ((-pow(P, -1, 1 << W)) % (1 << W), W))
print(arr("FQ_ONE29", limbs29(R261 % P)) + "  // R mod p, R = 2^%d" % RBITS)
print(arr("FQ_R2_29", limbs29(R261 * R261 % P)) + "  // R^2 mod p")
print(arr("FQ_2P266_29", limbs29(pow(2, 2 * RBITS - 64 * HL, P))) + "  // 2^%d mod p: (x * 2^%d) -> (x * 2^%d), i.e. R = 2^%d Montgomery input (%d x 64-bit limbs) to device form" % (2 * RBITS - 64 * HL, 64 * HL, RBITS, 64 * HL, HL))
print(arr("FQ_B29", limbs29(CURVE_B * R261 % P)) + "  // curve constant b in Montgomery form")
# square roots for the point sampler: p = 3 mod 4 -> a^((p+1)/4); otherwise Tonelli-Shanks with p - 1 = 2^s t
if P % 4 == 3:
    sq_s, sq_t, sq_c0 = 0, 0, 0
else:
    sq_t, sq_s = P - 1, 0
    while sq_t % 2 == 0:
        sq_t //= 2
        sq_s += 1
    z = 2
    while pow(z, (P - 1) // 2, P) == 1:
        z += 1
    sq_c0 = pow(z, sq_t, P)
print("constexpr int FQ_SQRT_S = %d;  // 2-adicity of p - 1 (0: p = 3 mod 4, the sampler uses a^((p+1)/4))" % sq_s)
print(arr("FQ_SQRT_T_32", limbs(sq_t, 32, WORDS)) + "  // t: p - 1 = 2^s t")
print(arr("FQ_SQRT_TP1H_32", limbs((sq_t + 1) // 2, 32, WORDS)) + "  // (t + 1) / 2")
print(arr("FQ_SQRT_C0_29", limbs29(sq_c0 * R261 % P)) + "  // z^t for the smallest non-residue z, Montgomery form")
print("// k*p in borrow-proof form (limbs 0..7 >= 2^31 - 4): a + KP[k] - b is limb-wise non-negative for b < (k-1)*p")
print("static constexpr uint32_t FQ_KP29[17][%d] = {" % L)
for k in range(17):
    vals = borrow_proof(k) if k >= 2 else [0] * L
    print("  {%s}," % ", ".join("0x%08xu" % v for v in vals))
print("};")
# 2p with limbs 0..7 in [2^29 - 1, 2^30): l_i - y_i needs no borrow for an exact y (limbs < 2^29) and stays a lazy limb (< 2^30)
q2 = limbs29(2 * P)
lazy2p = [q2[0] + (1 << W)] + [q2[i] + (1 << W) - 1 for i in range(1, L - 1)] + [q2[L - 1] - 1]
assert sum(v << (W * i) for i, v in enumerate(lazy2p)) == 2 * P and all((1 << W) - 1 <= v < (1 << (W + 1)) for v in lazy2p[:L - 1])
assert lazy2p[L - 1] >= limbs29(P)[L - 1]
print(arr("FQ_2P_LAZY29", lazy2p) + "  // 2p, borrow-free against exact limbs: fq_neg_lazy")
print("// p as %d x 32-bit words (packed / canonical form)" % WORDS)
print(arr("FQ_P32", limbs(P, 32, WORDS)))
print("// host finalisation: %d x 64-bit limbs, R = 2^%d" % (HL, 64 * HL))
R256 = 1 << (64 * HL)
f64 = "0x%016xull"
print(arr("FQ_P64", limbs(P, 64, HL), "uint64_t", f64))
print("constexpr uint64_t FQ_N0_64 = 0x%016xull;  // -p^-1 mod 2^64" % ((-pow(P, -1, 1 << 64)) % (1 << 64)))
print(arr("FQ_ONE64", limbs(R256 % P, 64, HL), "uint64_t", f64) + "  // 2^%d mod p" % (64 * HL))
print(arr("FQ_R2_64", limbs(R256 * R256 % P, 64, HL), "uint64_t", f64) + "  // 2^%d mod p" % (128 * HL))
print(arr("FR_R32", limbs(RMOD, 32, 8)) + "  // scalar field modulus r (for the samplers' rejection test)")
print(arr("FR_R29", limbs29(RMOD)) + "  // r in the unit's limbs (scalars handed over in R = 2^256 Montgomery form)")
print("static constexpr uint32_t FR_N0_29 = 0x%08xu;  // -r^-1 mod 2^%d" % ((-pow(RMOD, -1, 1 << W)) % (1 << W), W))
print(arr("FQ_PP1D4_32", limbs((P + 1) // 4, 32, WORDS)) + "  // (p + 1) / 4 : square-root exponent")
print(arr("FQ_PM2_32", limbs(P - 2, 32, WORDS)) + "  // p - 2 : inversion exponent")
print("constexpr int FQ_BITS = %d;  // bit length of p (exponent loops)" % P.bit_length())


# ---- the curve endomorphism phi(x, y) = (beta x, y) = lambda (x, y) (both curves have j = 0) and the scalar split
# k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^127 (Gallant-Lambert-Vanstone; csrc/glv.h states the arithmetic)
def ec_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def ec_mul(k, pt):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, pt)
        pt = ec_add(pt, pt)
        k >>= 1
    return acc


BETA = cube_root_of_unity(P)
LAMBDA = cube_root_of_unity(RMOD)
if ec_mul(LAMBDA, GEN) != (BETA * GEN[0] % P, GEN[1]):
    LAMBDA = LAMBDA * LAMBDA % RMOD
assert ec_mul(LAMBDA, GEN) == (BETA * GEN[0] % P, GEN[1])
_glv = glv_lattice(RMOD, LAMBDA)
V1, V2, G1, G2, N11, N12, N21, N22, GLV_OK = (_glv[k] for k in ("V1", "V2", "G1", "G2", "N11", "N12", "N21", "N22", "ok"))
GLV_SHIFT = 320
assert GLV_OK or CURVE == "bls12_381", "the halves do not fit 8 signed 16-bit windows"
if not GLV_OK:  # BLS12-381: the 255-bit r splits into halves of up to 128 bits, one more than 8 signed 16-bit windows hold: no endomorphism mode
    G1 = G2 = N11 = N12 = N21 = N22 = 0
print("constexpr bool GLV_SUPPORTED = %s;  // MSM_HIP_BASES_ENDOMORPHISM available for this curve" % ("true" if GLV_OK else "false"))
print("// endomorphism phi(x, y) = (beta x, y) = lambda (x, y); k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^127: csrc/glv.h")
print("// lambda = %d" % LAMBDA)
print("// lattice basis (a1, b1) = (%d, %d), (a2, b2) = (%d, %d)" % (V1[0], V1[1], V2[0], V2[1]))
print(arr("FQ_BETA29", limbs29(BETA * R261 % P)) + "  // beta (a primitive cube root of unity mod p), Montgomery form")
print("constexpr int GLV_SHIFT = %d;" % GLV_SHIFT)
print(arr("GLV_G1_32", limbs(G1, 32, 7)) + "  // round(2^320 |b2| / r)")
print(arr("GLV_G2_32", limbs(G2, 32, 7)) + "  // round(2^320 |b1| / r)")
print(arr("GLV_N11_32", limbs(N11, 32, 5)) + "  // -sign(b2) a1 mod 2^160")
print(arr("GLV_N12_32", limbs(N12, 32, 5)) + "  // -sign(-b1) a2 mod 2^160")
print(arr("GLV_N21_32", limbs(N21, 32, 5)) + "  // -sign(b2) b1 mod 2^160")
print(arr("GLV_N22_32", limbs(N22, 32, 5)) + "  // -sign(-b1) b2 mod 2^160")
print("}  // namespace MSM_FIELD_NS")
