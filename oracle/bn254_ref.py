"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Pure-Python big-integer model of the BN254 G1 MSM hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.

This is the slow, trivially auditable half of the oracle: affine formulas over Python ints with
`pow(x, -1, p)`.  It exists to (a) pin the fast C restatement (`oracle/bn254.c`) and (b) generate the
golden vectors under `tests/golden/` (see `tests/golden/make_golden.py`).

Parity status: the reference (`/root/reference`, Rust + WGSL) cannot be built or
imported here and holds NO golden MSM vectors (all its tests draw from an unseeded `thread_rng()`,
`src/lib.rs:21,27,37`).  MSM-output parity is therefore *unpinned by reference data*; it is pinned
mathematically -- an MSM result is a unique group element, so its canonical 64-byte affine encoding is
implementation independent -- plus by every known-answer constant the reference does hold
(`src/cuzk/utils.rs:439-451`, `src/naive/utils/bigint.rs:83-93`, `src/cuzk/msm.rs:39`), which
`tests/test_oracle.py` checks.

Stage semantics restated from (all paths relative to /root/reference):
  wire format ............ src/lib.rs:50-65, src/cuzk/utils.rs:10-21
  signed digit recode .... src/cuzk/wgsl/cuzk/decompose_scalars.template.wgsl:83-112, src/cuzk/test/utils.rs:121-161
  transpose (CSC build) .. src/cuzk/wgsl/cuzk/transpose.template.wgsl:47-73, src/cuzk/test/utils.rs:61-118
  SMVP ................... src/cuzk/wgsl/cuzk/smvp.template.wgsl:44-114, src/cuzk/test/utils.rs:166-219
  bucket reduction ....... src/cuzk/wgsl/cuzk/bpr.template.wgsl:38-132, src/cuzk/test/utils.rs:222-338
  window combine ......... src/cuzk/msm.rs:391-416
"""

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # src/cuzk/msm.rs:39
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # src/naive/utils/bigint.rs:85
B = 3                     # y^2 = x^3 + 3
G = (1, 2)                # generator, SURVEY.md Appendix A.8
INF = None                # affine point at infinity

MASK64 = (1 << 64) - 1
CB = 32  # bytes of a coordinate on the wire (48 in the BLS12-381 instance of this model, oracle/bls12_381_ref.py); scalars: 32 everywhere


# ---------------------------------------------------------------------------------------------------
# curve arithmetic (affine, big ints)
# ---------------------------------------------------------------------------------------------------
def is_on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return (y * y - (x * x * x + B)) % P == 0


def neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def add(p1, p2):
    """Complete affine addition; handles identity, doubling and inverse pairs
    (the same case split as src/cuzk/wgsl/curve/ec.template.wgsl:36-65)."""
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
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
    """sum_i s_i * P_i by independent double-and-add; the definition every other path must equal."""
    acc = INF
    for pt, s in zip(points, scalars):
        acc = add(acc, mul(s, pt))
    return acc


# ---------------------------------------------------------------------------------------------------
# wire format  (src/lib.rs:50-65, src/cuzk/utils.rs:10-21): canonical little-endian, 32 B per element
# ---------------------------------------------------------------------------------------------------
def points_to_bytes(points):
    out = bytearray()
    for pt in points:
        if pt is INF:
            raise ValueError("point at infinity is not representable (src/lib.rs:58 panics)")
        out += pt[0].to_bytes(CB, "little") + pt[1].to_bytes(CB, "little")
    return bytes(out)


def scalars_to_bytes(scalars):
    return b"".join((s % R).to_bytes(32, "little") for s in scalars)


def bytes_to_points(b):
    assert len(b) % (2 * CB) == 0
    return [(int.from_bytes(b[i:i + CB], "little"), int.from_bytes(b[i + CB:i + 2 * CB], "little"))
            for i in range(0, len(b), 2 * CB)]


def bytes_to_scalars(b):
    assert len(b) % 32 == 0
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def jacobian_bytes_to_affine(xyz):
    """96 B x||y||z canonical LE (z = 0 => infinity) -> affine tuple or INF."""
    x = int.from_bytes(xyz[0:CB], "little")
    y = int.from_bytes(xyz[CB:2 * CB], "little")
    z = int.from_bytes(xyz[2 * CB:3 * CB], "little")
    if z == 0:
        return INF
    zi = pow(z, -1, P)
    return (x * zi * zi % P, y * zi * zi * zi % P)


def affine_to_bytes64(pt):
    """Canonical 64-byte affine encoding used for bit-exact comparison; infinity = 64 zero bytes."""
    if pt is INF:
        return bytes(2 * CB)
    return pt[0].to_bytes(CB, "little") + pt[1].to_bytes(CB, "little")


# ---------------------------------------------------------------------------------------------------
# limb codec known-answer helper (src/cuzk/utils.rs:24-50 `to_words_le`)
# ---------------------------------------------------------------------------------------------------
def to_words_le(val, num_words, word_size):
    mask = (1 << word_size) - 1
    return [(val >> (word_size * i)) & mask for i in range(num_words)]


# ---------------------------------------------------------------------------------------------------
# deterministic synthetic inputs (shared definition with oracle/bn254.c and the HIP sampler kernels)
#   splitmix64 counter hash; scalars by rejection below r; points by try-and-increment on x
#   (the reference's `sample_points` draws Curve::random = random x until x^3+3 is a square, src/lib.rs:36-42)
# ---------------------------------------------------------------------------------------------------
def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def _draw256(seed, index, attempt, domain, words64=4):
    """Four 64-bit words -> 254-bit candidate (top two bits cleared); six words (a BLS12-381 coordinate) -> as many bits as P has."""
    base = splitmix64(seed ^ ((domain & 0xFF) << 56)) ^ ((index * 0xD1342543DE82EF95) & MASK64)
    base = splitmix64(base ^ ((attempt * 0xA0761D6478BD642F) & MASK64))
    v = 0
    s = base
    for k in range(words64):
        s = splitmix64(s)
        v |= s << (64 * k)
    return v & ((1 << (254 if words64 == 4 else P.bit_length())) - 1)


def sample_scalar(seed, index):
    attempt = 0
    while True:
        v = _draw256(seed, index, attempt, 1)
        if v < R:
            return v
        attempt += 1


def sqrt_mod(a):
    """A square root of a mod P, or None.  P = 3 mod 4 (BN254 Fq): a^((P+1)/4); otherwise Tonelli-Shanks (the Grumpkin
    instance of this model, oracle/grumpkin_ref.py: P - 1 = 2^28 t)."""
    a %= P
    if P % 4 == 3:
        y = pow(a, (P + 1) // 4, P)
        return y if y * y % P == a else None
    if a == 0:
        return 0
    if pow(a, (P - 1) // 2, P) != 1:
        return None
    t, s = P - 1, 0
    while t % 2 == 0:
        t //= 2
        s += 1
    z = 2
    while pow(z, (P - 1) // 2, P) == 1:
        z += 1
    x, b, c, m = pow(a, (t + 1) // 2, P), pow(a, t, P), pow(z, t, P), s
    while b != 1:
        i, q = 0, b
        while q != 1:
            q = q * q % P
            i += 1
        bb = pow(c, 1 << (m - i - 1), P)
        x, c = x * bb % P, bb * bb % P
        b, m = b * c % P, i
    return x


def sample_point(seed, index):
    attempt = 0
    while True:
        x = _draw256(seed, index, attempt, 2, CB // 8)
        if x < P:
            rhs = (x * x * x + B) % P
            y = sqrt_mod(rhs)
            if y is not None:
                if (y & 1) != ((x >> 1) & 1):        # pick the root by a data-dependent bit
                    y = P - y
                return (x, y)
        attempt += 1


def sample_scalars(seed, n):
    return [sample_scalar(seed, i) for i in range(n)]


def sample_points(seed, n):
    return [sample_point(seed, i) for i in range(n)]


# ---------------------------------------------------------------------------------------------------
# cuZK stage models (tiny sizes only)
# ---------------------------------------------------------------------------------------------------
def decompose_scalars_signed(scalars, num_words, word_size):
    """-> result[w][i] = biased digit d + 2^(c-1), window-major (src/cuzk/test/utils.rs:121-161)."""
    l = 1 << word_size
    h = l >> 1
    out = [[0] * len(scalars) for _ in range(num_words)]
    for i, s in enumerate(scalars):
        carry = 0
        for w in range(num_words):
            d = ((s >> (word_size * w)) & (l - 1)) + carry
            if d >= h:
                d -= l
                carry = 1
            else:
                carry = 0
            out[w][i] = d + h
        if carry:
            raise ValueError("final carry is 1 (src/cuzk/test/utils.rs:150-152)")
    return out


def cpu_transpose(digits_w, num_columns):
    """One window: -> (col_ptr[num_columns+1], val_idxs[n]); stable counting sort of point indices
    by biased digit (src/cuzk/test/utils.rs:61-118)."""
    n = len(digits_w)
    col_ptr = [0] * (num_columns + 1)
    for d in digits_w:
        col_ptr[d + 1] += 1
    for b in range(num_columns):
        col_ptr[b + 1] += col_ptr[b]
    cur = [0] * num_columns
    val = [0] * n
    for i, d in enumerate(digits_w):
        val[col_ptr[d] + cur[d]] = i
        cur[d] += 1
    return col_ptr, val


def cpu_smvp_signed(col_ptr, val_idxs, points, num_columns):
    """One window: -> buckets[h]; slot k>=1: sum(d=+k) - sum(d=-k); slot 0: -sum(d=-h)
    (src/cuzk/test/utils.rs:166-219)."""
    h = num_columns // 2
    buckets = [INF] * h
    for k in range(h):
        for j in range(2):
            row = k + h if j == 0 else h - k
            if k == 0 and j == 0:
                row = 0
            s = INF
            for t in range(col_ptr[row], col_ptr[row + 1]):
                s = add(s, points[val_idxs[t]])
            if h > row:
                bi = h - row
                s = neg(s)
            else:
                bi = row - h
            if bi > 0:
                buckets[k] = add(buckets[k], s)
    return buckets


def serial_bucket_reduction(buckets):
    """sum_{k>=1} k*B[k] + h*B[0]  (src/cuzk/test/utils.rs:222-235)."""
    h = len(buckets)
    acc = INF
    for k in range(1, h):
        acc = add(acc, mul(k, buckets[k]))
    return add(acc, mul(h, buckets[0]))


def running_sum_bucket_reduction(buckets):
    """src/cuzk/test/utils.rs:238-251."""
    n = len(buckets)
    m = buckets[0]
    g = m
    for i in range(n - 1):
        m = add(m, buckets[n - 1 - i])
        g = add(g, m)
    return g


def parallel_bucket_reduction(buckets, num_threads):
    """src/cuzk/test/utils.rs:255-284 (the two-stage GPU split, bpr.template.wgsl:38-132, fused)."""
    per = len(buckets) // num_threads
    outs = []
    for t in range(num_threads):
        idx = 0 if t == 0 else (num_threads - t) * per
        m = buckets[idx]
        g = m
        for i in range(per - 1):
            m = add(m, buckets[(num_threads - t) * per - 1 - i])
            g = add(g, m)
        s = per * (num_threads - t - 1)
        if s > 0:
            g = add(g, mul(s, m))
        outs.append(g)
    return outs


def horner(window_sums, word_size):
    """result = sum_w 2^(c*w) * S_w from the top window down (src/cuzk/msm.rs:411-416)."""
    acc = window_sums[-1]
    for w in range(len(window_sums) - 2, -1, -1):
        acc = add(mul(1 << word_size, acc), window_sums[w])
    return acc


def msm_cuzk_model(points, scalars, word_size=16):
    """The whole reference pipeline on CPU models, cf. tests/cuzk.rs:11-95 (with correct, unpadded indexing)."""
    num_words = -(-256 // word_size)
    num_columns = 1 << word_size
    digits = decompose_scalars_signed(scalars, num_words, word_size)
    sums = []
    for w in range(num_words):
        col_ptr, val = cpu_transpose(digits[w], num_columns)
        buckets = cpu_smvp_signed(col_ptr, val, points, num_columns)
        sums.append(running_sum_bucket_reduction(buckets))
    return horner(sums, word_size)


# ---------------------------------------------------------------------------------------------------
# curve endomorphism ("GLV endomorphism ... to halve scalar length"; no counterpart in the reference, which
# uses full-length scalars, src/cuzk/msm.rs:79-82).  phi(x, y) = (beta x, y) = lambda (x, y); k = k1 + k2 lambda (mod r).
# Derived here from P, R, G alone, independently of tools/gen_constants.py; tests compare the two.
# ---------------------------------------------------------------------------------------------------
GLV_SHIFT = 320
_glv_cache = {}


def _cube_root_of_unity(mod):
    g = 2
    while pow(g, (mod - 1) // 3, mod) == 1:
        g += 1
    return pow(g, (mod - 1) // 3, mod)


def glv_params():
    """-> dict(beta, lam, v1, v2, g1, g2, s1, s2): beta / lambda paired on the generator; (v1, v2) the short lattice basis from
    the extended Euclidean sequence of (r, lambda) around sqrt(r) with det = +r; g_i = round(2^320 |b| / r) and the signs of
    c1 = round(k b2 / r), c2 = round(-k b1 / r)."""
    key = (P, R, G)
    if key in _glv_cache:
        return _glv_cache[key]
    beta = _cube_root_of_unity(P)
    lam = _cube_root_of_unity(R)
    if mul(lam, G) != (beta * G[0] % P, G[1]):
        lam = lam * lam % R
    assert mul(lam, G) == (beta * G[0] % P, G[1])
    seq = [(R, 0), (lam, 1)]
    while seq[-1][0]:
        q = seq[-2][0] // seq[-1][0]
        seq.append((seq[-2][0] - q * seq[-1][0], seq[-2][1] - q * seq[-1][1]))
    l = max(i for i, (rem, _) in enumerate(seq) if rem * rem >= R)
    v1 = (seq[l + 1][0], -seq[l + 1][1])
    v2 = min([(seq[l][0], -seq[l][1]), (seq[l + 2][0], -seq[l + 2][1])], key=lambda v: v[0] * v[0] + v[1] * v[1])
    if v1[0] * v2[1] - v2[0] * v1[1] < 0:
        v2 = (-v2[0], -v2[1])
    assert v1[0] * v2[1] - v2[0] * v1[1] == R
    out = {"beta": beta, "lam": lam, "v1": v1, "v2": v2,
           "g1": ((abs(v2[1]) << GLV_SHIFT) + R // 2) // R, "g2": ((abs(v1[1]) << GLV_SHIFT) + R // 2) // R,
           "s1": 1 if v2[1] >= 0 else -1, "s2": 1 if -v1[1] >= 0 else -1}
    _glv_cache[key] = out
    return out


def glv_split(k):
    """-> (k1, k2), signed, with k = k1 + k2 lambda (mod r) -- the rounding is the engine's (csrc/glv.h): (k g + 2^319) >> 320."""
    q = glv_params()
    c1 = q["s1"] * ((k * q["g1"] + (1 << (GLV_SHIFT - 1))) >> GLV_SHIFT)
    c2 = q["s2"] * ((k * q["g2"] + (1 << (GLV_SHIFT - 1))) >> GLV_SHIFT)
    return k - c1 * q["v1"][0] - c2 * q["v2"][0], -c1 * q["v1"][1] - c2 * q["v2"][1]


def endo(pt):
    """phi(P) = (beta x, y) = lambda P"""
    return None if pt is None else (glv_params()["beta"] * pt[0] % P, pt[1])
