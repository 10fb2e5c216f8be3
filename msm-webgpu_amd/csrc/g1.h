// BN254 G1 (y^2 = x^3 + 3) point arithmetic on fq29 limbs, extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2).
//
// The reference adds full Jacobian points with add-2007-bl at 11M + 5S even when the incoming point is affine
// (src/cuzk/wgsl/curve/ec.template.wgsl:36-86, used by smvp.template.wgsl:70-79).  The bucket accumulator here
// adds an affine point into an XYZZ accumulator (EFD madd-2008-s, 8M + 2S) and full XYZZ additions
// (add-2008-s, 12M + 2S) / doublings (dbl-2008-s-1) are used by the bucket reduction.  The case split is the
// reference's (ec.template.wgsl:36-65): P = inf, Q = inf, P = Q -> double, P = -Q -> inf.
//
// Value bounds kept between operations (fq29.h conventions; all coordinates "normal"):
//     X < 9p,  Y < 5p,  ZZ < 2p (exact),  ZZZ < 2p (exact)
// Every fq_sub<K> below states the bound of its subtrahend in the trailing comment.
// The products' operand bounds are written in units of p^2 beside each multiplication; a Montgomery product needs
// value(a) value(b) <= FQ_HEADROOM p^2 (169 for BN254 / Grumpkin, 127 for the 255-bit Pallas / Vesta moduli) to return < 2p.  The
// comments use the loose "< 2p" for every product's output; the one place where that is not enough for FQ_HEADROOM = 127 is P^2 in the
// mixed additions ("144 p^2" with P < 12p).  The tight bound holds for every modulus: a product of operands with value(a) value(b) <= K p^2
// is < (1 + K / FQ_HEADROOM) p, the stored bases are such products with K <= 1 (fq_to_mont, the endomorphism's beta x) or canonical,
// ZZ / ZZZ are products with K <= 4, so U2 = px ZZ < 1.01 p, P = U2 - X + 10p < 11.02 p and P^2 <= 121.5 p^2 < 127 p^2.
// tests/test_fq29_host.py builds these formulas for all four curves with every product's result asserted below 2p, on random chains
// and with the inputs raised to the edge of the bounds above.
#ifndef MSM_CURVE_UNIT
#pragma once
#include "fq29.h"
#endif

// The full addition and the doubling are inlined into every caller (G1_HD = FQ_HD) except in a unit that asks for calls instead
// (MSM_G1_OUTLINE: BLS12-381 G2, where one inlined addition is ~16 000 instructions and the reduce kernels hold several: the unit's compile
// time, not its speed, is what the calls are for; the SMVP's mixed addition stays inline everywhere).
#if defined(MSM_G1_OUTLINE) && defined(__HIPCC__)
#define G1_HD __host__ __device__ __noinline__
#else
#define G1_HD FQ_HD
#endif

namespace MSM_FIELD_NS {

struct g1_affine {  // Montgomery form, canonical
  fq x, y;
};

struct g1_xyzz {
  fq x, y, zz, zzz;
  bool inf;
};

FQ_HD g1_xyzz g1_identity() {
  g1_xyzz r;
  r.x = fq_zero();
  r.y = fq_zero();
  r.zz = fq_zero();
  r.zzz = fq_zero();
  r.inf = true;
  return r;
}

FQ_HD g1_xyzz g1_from_affine(const fq& px, const fq& py) {
  g1_xyzz r;
  r.x = px;
  r.y = py;
  r.zz = fq_one();
  r.zzz = fq_one();
  r.inf = false;
  return r;
}

// 2 * (px, py) for an affine point (EFD mdbl-2008-s-1)
FQ_HD g1_xyzz g1_double_affine(const fq& px, const fq& py) {
  g1_xyzz r;
  const fq U = fq_dbl(py);                              // < 2p, lazy
  const fq V = fq_sqr(U);                               // < 2p
  const fq W = fq_mul(U, V);                            // < 2p
  const fq S = fq_mul(px, V);                           // < 2p
  const fq XX = fq_sqr(px);                             // < 2p
  const fq M = fq_norm(fq_add(fq_dbl(XX), XX));         // < 6p
  const fq MM = fq_sqr(M);                              // 36 p^2
  r.x = fq_sub<5>(MM, fq_dbl(S));                       // 2S < 4p      -> X < 7p
  const fq T = fq_sub<8>(S, r.x);                       // X < 7p       -> T < 10p
  r.y = fq_sub<3>(fq_mul(M, T), fq_mul(W, py));         // < 2p         -> Y < 5p
  r.zz = V;
  r.zzz = W;
  r.inf = false;
  return r;
}

// 2 * a  (EFD dbl-2008-s-1, a = 0)
G1_HD g1_xyzz g1_double(const g1_xyzz& a) {
  if (a.inf) return a;
  g1_xyzz r;
  const fq U = fq_dbl(a.y);                             // < 10p, lazy
  const fq V = fq_sqr(U);                               // 100 p^2
  const fq W = fq_mul(U, V);                            // 20 p^2
  const fq S = fq_mul(a.x, V);                          // 18 p^2
  const fq XX = fq_sqr(a.x);                            // 81 p^2
  const fq M = fq_norm(fq_add(fq_dbl(XX), XX));         // < 6p
  const fq MM = fq_sqr(M);
  r.x = fq_sub<5>(MM, fq_dbl(S));                       // 2S < 4p      -> X < 7p
  const fq T = fq_sub<8>(S, r.x);                       // X < 7p       -> T < 10p
  r.y = fq_sub<3>(fq_mul(M, T), fq_mul(W, a.y));        // < 2p         -> Y < 5p
  r.zz = fq_mul(V, a.zz);
  r.zzz = fq_mul(W, a.zzz);
  r.inf = false;
  return r;
}

// a += (px, py)   mixed addition, the SMVP inner operation (EFD madd-2008-s: 8M + 2S)
FQ_HD void g1_madd(g1_xyzz& a, const fq& px, const fq& py) {
  if (a.inf) {
    a = g1_from_affine(px, py);
    return;
  }
  const fq U2 = fq_mul_fast(px, a.zz);                       // < 2p
  const fq S2 = fq_mul_fast(py, a.zzz);                      // < 2p
  const fq P = fq_sub<10>(U2, a.x);                     // X < 9p       -> P < 12p
  const fq nY = fq_sub<6>(fq_zero(), a.y);              // Y < 5p       -> -Y < 6p, normal
  const fq R = fq_add(S2, nY);                          // S2 - Y + 6p < 8p, lazy limbs (no second subtraction / carry pass)
  const fq PP = fq_sqr_fast(P);                              // 144 p^2
  // same x (P = Q or P = -Q) <=> PP = 0 mod p, i.e. the exact limbs are all 0 or equal p's: the lowest limb filters first,
  // so the full comparison is almost never executed
  if ((PP.v[0] == 0u || PP.v[0] == FQ_P29[0]) && fq_is_zero_exact(PP)) {
    if (fq_is_zero_exact(fq_tidy(R)))
      a = g1_double_affine(px, py);
    else
      a = g1_identity();
    return;
  }
  const fq PPP = fq_mul_fast(P, PP);                         // 24 p^2
  const fq Q = fq_mul_fast(a.x, PP);                         // 18 p^2
  const fq RR = fq_sqr_fast(R);                              // 64 p^2
  const fq X3 = fq_sub<7>(RR, fq_add(PPP, fq_dbl(Q)));  // PPP + 2Q < 6p -> X3 < 9p
  const fq T = fq_sub<10>(Q, X3);                       // X3 < 9p      -> T < 12p
  a.y = fq_mul2_fast(R, T, nY, PPP);                    // 96 + 12 p^2, one reduction -> Y3 < 2p (R lazy, the others normal)
  a.x = X3;
  a.zz = fq_mul_fast(a.zz, PP);
  a.zzz = fq_mul_fast(a.zzz, PPP);
}

// The SMVP's form of the mixed addition: the accumulator keeps W = (wneg ? -Y : Y) instead of Y.
//   g1_madd needs -Y twice per step (R = S2 - Y and Y3 = R T - Y PPP): a subtraction with a carry pass.  With R' = W + (-sigma s py) ZZZ
//   = -sigma R (sigma = +-1 the sign W carries, s = +-1 the digit's sign: the point's y is negated anyway, now by sigma s instead
//   of s, and lazily: fq_neg_lazy) the same products give  R'^2 = R^2  and  R' T + W PPP = -sigma (R T - Y PPP) = -sigma Y3:
//   the new W, with the opposite sign.  No negation of the accumulator at all; the sign bit flips every step and is applied once,
//   when the accumulator is flushed (g1_unsigned).  Saves ~60 of the ~2270 VALU instructions of a step.
// `sneg`: the digit is negative (the point enters as -P).
FQ_HD void g1_madd_w(g1_xyzz& a, bool& wneg, const fq& px, const fq& py, bool sneg) {
  if (a.inf) {  // the first point of a run: W = py with the digit's sign as the state -- no negation (zz, zzz of an empty accumulator are don't-care)
    a.x = px;
    a.y = py;
    a.zz = fq_one();
    a.zzz = fq_one();
    a.inf = false;
    wneg = sneg;
    return;
  }
  const bool negp = sneg == wneg;                        // -sigma s = -1
  fq pye;                                               // (-sigma s) py: py itself or 2p - py (lazy limbs < 2^30)
  {
    const fq n = fq_neg_lazy(py);
#pragma unroll
    for (int i = 0; i < FQ_L; i++) pye.v[i] = negp ? n.v[i] : py.v[i];
  }
  const fq U2 = fq_mul_fast(px, a.zz);                  // < 2p
  const fq S2 = fq_mul_fast(pye, a.zzz);                // 2p * 2p           -> < 2p, exact
  const fq P = fq_sub<10>(U2, a.x);                     // X < 9p            -> P < 12p
  const fq Rw = fq_add(a.y, S2);                        // W < 5p (normal)   -> R' < 7p, lazy limbs
  const fq PP = fq_sqr_fast(P);                         // 144 p^2
  if ((PP.v[0] == 0u || PP.v[0] == FQ_P29[0]) && fq_is_zero_exact(PP)) {  // same x: P = Q or P = -Q (see g1_madd)
    if (fq_is_zero_exact(fq_tidy(Rw)))
      a = g1_double_affine(px, sneg ? fq_neg_canonical(py) : py);
    else
      a = g1_identity();
    wneg = false;
    return;
  }
  const fq PPP = fq_mul_fast(P, PP);                    // 24 p^2
  const fq Q = fq_mul_fast(a.x, PP);                    // 18 p^2
  const fq RR = fq_sqr_fast(Rw);                        // 49 p^2
  const fq X3 = fq_sub<7>(RR, fq_add(PPP, fq_dbl(Q)));  // PPP + 2Q < 6p     -> X3 < 9p
  const fq T = fq_sub<10>(Q, X3);                       // X3 < 9p           -> T < 12p
  // the three products that update a loop-carried coordinate write their result over it (fq_mul_fast_ip: no copy back)
  fq_mul2_fast_ip(Rw, T, a.y, PPP);                     // 84 + 10 p^2, one reduction -> -sigma Y3 < 2p (R' lazy, the others normal)
  a.x = X3;
  fq_mul_fast_ip(a.zz, PP);
  fq_mul_fast_ip(a.zzz, PPP);
  wneg = !wneg;
}
// The SMVP loop's form of g1_madd_w: `a` is not empty.  Every product is computed unconditionally and the loop-carried coordinates are
// updated in place, so the hot path has no merge of alternative results (each merge costs a register copy per limb).  Returns 0, or --
// same x coordinate, the products above are void -- 1: the sum is 2 (+-P), 2: the sum is the identity; the caller repairs `a` (rare).
FQ_HD int g1_madd_w_hot(g1_xyzz& a, bool& wneg, const fq& px, const fq& py, bool sneg) {
  const bool negp = sneg == wneg;                        // -sigma s = -1
  fq pye;                                               // (-sigma s) py: py itself or 2p - py (lazy limbs < 2^30)
  {
    const fq n = fq_neg_lazy(py);
#pragma unroll
    for (int i = 0; i < FQ_L; i++) pye.v[i] = negp ? n.v[i] : py.v[i];
  }
  const fq U2 = fq_mul_fast(px, a.zz);                  // < 2p
  const fq S2 = fq_mul_fast(pye, a.zzz);                // 2p * 2p           -> < 2p, exact
  const fq P = fq_sub<10>(U2, a.x);                     // X < 9p            -> P < 12p
  const fq Rw = fq_add(a.y, S2);                        // W < 5p (normal)   -> R' < 7p, lazy limbs
  const fq PP = fq_sqr_fast(P);                         // 144 p^2
  int status = 0;
  if ((PP.v[0] == 0u || PP.v[0] == FQ_P29[0]) && fq_is_zero_exact(PP))  // same x: P = Q or P = -Q (the lowest limb filters first)
    status = fq_is_zero_exact(fq_tidy(Rw)) ? 1 : 2;
  const fq PPP = fq_mul_fast(P, PP);                    // 24 p^2
  const fq Q = fq_mul_fast(a.x, PP);                    // 18 p^2
  const fq RR = fq_sqr_fast(Rw);                        // 49 p^2
  const fq X3 = fq_sub<7>(RR, fq_add(PPP, fq_dbl(Q)));  // PPP + 2Q < 6p     -> X3 < 9p
  const fq T = fq_sub<10>(Q, X3);                       // X3 < 9p           -> T < 12p
  fq_mul2_fast_ip(Rw, T, a.y, PPP);                     // 84 + 10 p^2, one reduction -> -sigma Y3 < 2p
  a.x = X3;
  fq_mul_fast_ip(a.zz, PP);
  fq_mul_fast_ip(a.zzz, PPP);
  wneg = !wneg;
  return status;
}
// the accumulator of g1_madd_w with its sign applied: a plain XYZZ point (Y < 5p, normal)
FQ_HD g1_xyzz g1_unsigned(const g1_xyzz& a, bool wneg) {
  g1_xyzz r = a;
  if (wneg && !a.inf) r.y = fq_sub<3>(fq_zero(), a.y);  // wneg only with W a product's result or a canonical y: W < 2p -> -W < 3p
  return r;
}

// a + b   (EFD add-2008-s: 12M + 2S)
G1_HD g1_xyzz g1_add(const g1_xyzz& a, const g1_xyzz& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  const fq U1 = fq_mul(a.x, b.zz);                      // 18 p^2
  const fq U2 = fq_mul(b.x, a.zz);
  const fq S1 = fq_mul(a.y, b.zzz);                     // 10 p^2
  const fq S2 = fq_mul(b.y, a.zzz);
  const fq P = fq_sub<3>(U2, U1);                       // < 2p         -> P < 5p
  const fq nS1 = fq_sub<3>(fq_zero(), S1);              // S1 < 2p      -> -S1 < 3p, normal
  const fq R = fq_add(S2, nS1);                         // S2 - S1 + 3p < 5p, lazy limbs
  const fq PP = fq_sqr(P);
  if ((PP.v[0] == 0u || PP.v[0] == FQ_P29[0]) && fq_is_zero_exact(PP)) {  // lowest limb filters first (see g1_madd)
    if (fq_is_zero_exact(fq_tidy(R))) return g1_double(a);
    return g1_identity();
  }
  g1_xyzz r;
  const fq PPP = fq_mul(P, PP);
  const fq Q = fq_mul(U1, PP);
  const fq RR = fq_sqr(R);
  r.x = fq_sub<7>(RR, fq_add(PPP, fq_dbl(Q)));          // < 6p         -> X3 < 9p
  const fq T = fq_sub<10>(Q, r.x);                      // X3 < 9p      -> T < 12p
  r.y = fq_mul2(R, T, nS1, PPP);                        // 60 + 6 p^2, one reduction -> Y3 < 2p (R lazy, the others normal)
  r.zz = fq_mul(fq_mul(a.zz, b.zz), PP);
  r.zzz = fq_mul(fq_mul(a.zzz, b.zzz), PPP);
  r.inf = false;
  return r;
}

// k * a for a small scalar, left-to-right double-and-add (≙ double_and_add, ec.template.wgsl:124-139)
FQ_HD g1_xyzz g1_mul_u32(const g1_xyzz& a, uint32_t k) {
  g1_xyzz acc = g1_identity();
  for (int bit = 31; bit >= 0; bit--) {
    acc = g1_double(acc);
    if ((k >> bit) & 1u) acc = g1_add(acc, a);
  }
  return acc;
}

// XYZZ -> Jacobian (X', Y', Z') with Z' = ZZ:  X' = X*ZZ, Y' = Y*ZZZ  (x = X'/Z'^2, y = Y'/Z'^3).
// Outputs Montgomery-form canonical values; the identity is (0, 0, 0) -- Z = 0 <=> identity, as ec.template.wgsl:4-8.
FQ_HD void g1_to_jacobian(const g1_xyzz& a, fq& X, fq& Y, fq& Z) {
  if (a.inf) {
    X = fq_zero();
    Y = fq_zero();
    Z = fq_zero();
    return;
  }
  X = fq_canonical(fq_mul(a.x, a.zz));
  Y = fq_canonical(fq_mul(a.y, a.zzz));
  Z = fq_canonical(a.zz);
}

// Jacobian (Montgomery, canonical) -> XYZZ: ZZ = Z^2, ZZZ = Z^3
FQ_HD g1_xyzz g1_from_jacobian(const fq& X, const fq& Y, const fq& Z) {
  uint32_t z = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) z |= Z.v[i];
  if (z == 0) return g1_identity();
  g1_xyzz r;
  r.x = X;
  r.y = Y;
  r.zz = fq_sqr(Z);
  r.zzz = fq_mul(r.zz, Z);
  r.inf = false;
  return r;
}

}  // namespace MSM_FIELD_NS
#undef G1_HD
