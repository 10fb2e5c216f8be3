"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Pure-Python big-integer model of MSM on BN254 G2 (SURVEY.md section 8f-4 "other curves / G2";
reference README.md future work: "Implement cuzk on other curves").

G2 is the order-r subgroup of the sextic twist  E': y^2 = x^3 + 3 / (9 + u)  over  Fq2 = Fq[u] / (u^2 + 1)  (the curve the pairing of
halo2curves::bn256 -- the reference's only curve, Cargo.toml:29 -- takes its second argument from).  The reference itself has no G2 code; this
model is the definition the HIP path is compared against, in two independent forms that are cross-checked in tests/test_oracle_g2.py:
`msm_naive` (affine double-and-add per point) and `msm_pippenger` (Jacobian coordinates, signed 16-bit windows as the reference's
decompose / bucket scheme, src/cuzk/wgsl/cuzk/decompose_scalars.template.wgsl:83-112).  Known answers it is pinned to: the generator of
EIP-197 lies on the curve and has order r; u^2 = -1; the twist constant times (9 + u) is 3.

Wire format (mirrors src/lib.rs:50-65 one level up the tower): an Fq2 element is c0 || c1, each 32 bytes canonical little-endian (64 B);
an affine point x || y (128 B); a Jacobian record x || y || z (192 B, z = 0 <=> infinity); scalars 32 B as for G1.
(oracle/bls12_381_g2_ref.py is this module with P, R, B, G, FB, CB and the G1 model it takes its scalars from rebound.)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package."""
from . import bn254_ref as _g1

P = _g1.P  # base field modulus, src/cuzk/msm.rs:39
R = _g1.R  # group order (the scalar field), src/naive/utils/bigint.rs:85
FB = 32    # bytes of an Fq element on the wire (48 in the BLS12-381 instance of this model, oracle/bls12_381_g2_ref.py)
CB = 64    # bytes of a coordinate (an Fq2 element c0 || c1) on the wire
INF = None


# --------------------------------------------------------------------------------------------------- Fq2 = Fq[u] / (u^2 + 1)
def f2(a, b=0):
    return (a % P, b % P)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)  # 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)
    return (a[0] * n % P, (-a[1]) * n % P)


ZERO, ONE = (0, 0), (1, 0)
B = f2_mul(f2(3), f2_inv(f2(9, 1)))  # the twist's constant 3 / (9 + u)
# generator of G2 (EIP-197 / the alt_bn128 precompile's P2; the same point as halo2curves::bn256::G2Affine::generator)
G = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531))


# --------------------------------------------------------------------------------------------------- curve arithmetic, affine
def is_on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return f2_sqr(y) == f2_add(f2_mul(f2_sqr(x), x), B)


def neg(pt):
    return INF if pt is INF else (pt[0], f2_neg(pt[1]))


def add(p1, p2):
    """Complete affine addition (identity, doubling and inverse pairs: the case split of src/cuzk/wgsl/curve/ec.template.wgsl:36-65)."""
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if f2_add(y1, y2) == ZERO:
            return INF
        lam = f2_mul(f2_scale(f2_sqr(x1), 3), f2_inv(f2_scale(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3)


def mul(k, pt):
    k %= R
    acc = INF
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def msm_naive(points, scalars):
    """sum_i s_i P_i by independent double-and-add: the definition."""
    acc = INF
    for pt, s in zip(points, scalars):
        acc = add(acc, mul(s, pt))
    return acc


# --------------------------------------------------------------------------------------------------- Jacobian (x = X / Z^2, y = Y / Z^3)
JINF = (ONE, ONE, ZERO)


def j_from_affine(pt):
    return JINF if pt is INF else (pt[0], pt[1], ONE)


def j_to_affine(p):
    X, Y, Z = p
    if Z == ZERO:
        return INF
    zi = f2_inv(Z)
    zi2 = f2_sqr(zi)
    return (f2_mul(X, zi2), f2_mul(Y, f2_mul(zi2, zi)))


def j_double(p):  # dbl-2009-l (a = 0), the formulas of ec.template.wgsl:67-86 over Fq2
    X, Y, Z = p
    if Z == ZERO:
        return p
    A, Bq = f2_sqr(X), f2_sqr(Y)
    C = f2_sqr(Bq)
    D = f2_scale(f2_sub(f2_sub(f2_sqr(f2_add(X, Bq)), A), C), 2)
    E = f2_scale(A, 3)
    X3 = f2_sub(f2_sqr(E), f2_scale(D, 2))
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), f2_scale(C, 8))
    return (X3, Y3, f2_scale(f2_mul(Y, Z), 2))


def j_add(p, q):  # add-2007-bl with the case split
    if p[2] == ZERO:
        return q
    if q[2] == ZERO:
        return p
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    Z1Z1, Z2Z2 = f2_sqr(Z1), f2_sqr(Z2)
    U1, U2 = f2_mul(X1, Z2Z2), f2_mul(X2, Z1Z1)
    S1, S2 = f2_mul(f2_mul(Y1, Z2), Z2Z2), f2_mul(f2_mul(Y2, Z1), Z1Z1)
    if U1 == U2:
        return j_double(p) if S1 == S2 else JINF
    H = f2_sub(U2, U1)
    I = f2_sqr(f2_scale(H, 2))
    J = f2_mul(H, I)
    r = f2_scale(f2_sub(S2, S1), 2)
    V = f2_mul(U1, I)
    X3 = f2_sub(f2_sub(f2_sqr(r), J), f2_scale(V, 2))
    Y3 = f2_sub(f2_mul(r, f2_sub(V, X3)), f2_scale(f2_mul(S1, J), 2))
    Z3 = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(Z1, Z2)), Z1Z1), Z2Z2), H)
    return (X3, Y3, Z3)


def signed_digits(s, c=16):
    """The reference's signed recode (decompose_scalars.template.wgsl:83-112): digits in [-2^(c-1), 2^(c-1))."""
    s %= R
    n = (254 + c) // c
    out, carry = [], 0
    for w in range(n):
        d = ((s >> (c * w)) & ((1 << c) - 1)) + carry
        carry = 0
        if d >= 1 << (c - 1):
            d -= 1 << c
            carry = 1
        out.append(d)
    assert carry == 0
    return out


def msm_pippenger(points, scalars, c=16):
    """Bucket method over Jacobian coordinates with signed c-bit windows: per window bucket sums, running-sum reduce, Horner combine
    (src/cuzk/msm.rs:391-416).  Shares nothing but the field helpers with msm_naive."""
    n = (254 + c) // c
    digs = [signed_digits(s, c) for s in scalars]
    jp = [j_from_affine(pt) for pt in points]
    jn = [j_from_affine(neg(pt)) for pt in points]
    acc = JINF
    for w in range(n - 1, -1, -1):
        for _ in range(c):
            acc = j_double(acc)
        buckets = {}
        for i in range(len(points)):
            d = digs[i][w]
            if d:
                k = abs(d)
                buckets[k] = j_add(buckets.get(k, JINF), jp[i] if d > 0 else jn[i])
        run, tot = JINF, JINF
        prev = None
        for k in sorted(buckets, reverse=True):  # sum_k k B_k by running sums with gaps: tot += (prev - k) * run between occupied buckets
            if prev is not None:
                tot = j_add(tot, j_to_jac_mul(run, prev - k))
            run = j_add(run, buckets[k])
            prev = k
        if prev is not None:
            tot = j_add(tot, j_to_jac_mul(run, prev))
        acc = j_add(acc, tot)
    return j_to_affine(acc)


def j_to_jac_mul(p, k):
    """k * p for a small non-negative integer k (double-and-add in Jacobian form)."""
    acc = JINF
    while k:
        if k & 1:
            acc = j_add(acc, p)
        p = j_double(p)
        k >>= 1
    return acc


# --------------------------------------------------------------------------------------------------- wire format
def f2_to_bytes(a):
    return a[0].to_bytes(FB, "little") + a[1].to_bytes(FB, "little")


def f2_from_bytes(b):
    return (int.from_bytes(b[0:FB], "little"), int.from_bytes(b[FB:2 * FB], "little"))


def points_to_bytes(points):
    out = bytearray()
    for pt in points:
        if pt is INF:
            raise ValueError("point at infinity is not representable (src/lib.rs:58 panics)")
        out += f2_to_bytes(pt[0]) + f2_to_bytes(pt[1])
    return bytes(out)


def bytes_to_points(b):
    assert len(b) % (2 * CB) == 0
    return [(f2_from_bytes(b[i:i + CB]), f2_from_bytes(b[i + CB:i + 2 * CB])) for i in range(0, len(b), 2 * CB)]


def scalars_to_bytes(scalars):
    return _g1.scalars_to_bytes(scalars)


def bytes_to_scalars(b):
    return _g1.bytes_to_scalars(b)


def sample_scalar(seed, index):
    return _g1.sample_scalar(seed, index)


def jacobian_bytes_to_affine(xyz):
    """192 B x || y || z (z = 0 => infinity) -> affine tuple or INF."""
    x, y, z = f2_from_bytes(xyz[0:CB]), f2_from_bytes(xyz[CB:2 * CB]), f2_from_bytes(xyz[2 * CB:3 * CB])
    for comp in x + y + z:
        assert comp < P, "non-canonical coordinate"
    return j_to_affine((x, y, z))


def affine_to_bytes(pt):
    """Canonical 128-byte affine encoding used for bit-exact comparison; infinity = 128 zero bytes."""
    return bytes(2 * CB) if pt is INF else f2_to_bytes(pt[0]) + f2_to_bytes(pt[1])


def sample_multipliers(n, seed=1):
    """m_i = a + i b (mod r) with P_i = m_i G for sample_points(n, seed): with them sum_i s_i P_i = (sum_i s_i m_i mod r) G in closed form --
    the expected result of an MSM of ANY size without running one (a size-independent check, independent of every bucket method)."""
    a = sample_scalar(seed ^ 0x6732, 0) | 1
    b = sample_scalar(seed ^ 0x6732, 1) | 1
    return [(a + i * b) % R for i in range(n)]


def msm_by_multipliers(multipliers, scalars):
    return mul(sum(m * s for m, s in zip(multipliers, scalars)) % R, G)


def sample_points(n, seed=1):
    """n distinct points of G2: P_i = (a + i b) G for seeded a, b (SURVEY.md 8d's synthetic points), by repeated addition in Jacobian
    form and one batched inversion."""
    a = sample_scalar(seed ^ 0x6732, 0) | 1
    b = sample_scalar(seed ^ 0x6732, 1) | 1
    cur, step = j_from_affine(mul(a, G)), j_from_affine(mul(b, G))
    jac = []
    for _ in range(n):
        jac.append(cur)
        cur = j_add(cur, step)
    # batch inversion of the z coordinates (Montgomery's trick)
    pref, run = [], ONE
    for p in jac:
        pref.append(run)
        run = f2_mul(run, p[2])
    inv = f2_inv(run)
    out = [None] * n
    for i in range(n - 1, -1, -1):
        zi = f2_mul(inv, pref[i])
        inv = f2_mul(inv, jac[i][2])
        zi2 = f2_sqr(zi)
        out[i] = (f2_mul(jac[i][0], zi2), f2_mul(jac[i][1], f2_mul(zi2, zi)))
    return out


# ---- the curve endomorphism on G2 (round 4; csrc/glv.h, tools/gen_constants.py emit_g2): the twist has j = 0, so phi(x, y) = (beta x, y) with
# beta a cube root of unity of the PRIME field is an endomorphism of the twist, and on the order-r subgroup it is the multiplication by the
# SAME lambda as on G1 (the split and its constants are the G1 model's) -- with the beta that belongs to lambda on G2
_glv_cache = {}


def glv_params():
    """-> the G1 model's dict (lam, v1, v2, g1, g2, s1, s2) with `beta` replaced by the cube root of unity mod p for which lambda G = (beta x_G, y_G) on G2"""
    key = (P, R, G)
    if key not in _glv_cache:
        q = dict(_g1.glv_params())
        lg = mul(q["lam"], G)
        b1 = q["beta"]
        q["beta"] = next(b for b in (b1, b1 * b1 % P) if lg == (f2_scale(G[0], b), G[1]))
        _glv_cache[key] = q
    return _glv_cache[key]


def glv_split(k):
    return _g1.glv_split(k)


def endo(pt):
    """phi(P) = (beta x, y) = lambda P for P of order r"""
    return None if pt is None else (f2_scale(pt[0], glv_params()["beta"]), pt[1])
