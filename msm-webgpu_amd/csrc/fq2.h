// The quadratic extension Fq2 = Fq[u] / (u^2 + 1) as the COORDINATE FIELD of a curve unit (BN254 G2: SURVEY.md 8f-4 "other curves / G2").
//
// g1.h, msm_kernels.h and host_g1.h are written against a field type `fq` and a fixed set of function names (fq_add, fq_sub<K>, fq_mul,
// fq_sqr, fq_mul2, the *_fast multipliers, fq_unpack / fq_pack, ...).  A G2 unit instantiates the prime field (fq29.h and its generated
// multipliers) in a namespace of its own, MSM_BASE_NS, and this header then defines the same names over Fq2 in the unit's namespace: an
// element is c0 + c1 u, stored as c0's limbs followed by c1's (FQ_L = 2 x 9 limbs; 16 packed words c0 || c1 on the wire and in the bases).
//
// Bounds.  The prime-field formulas of g1.h carry lazy values of up to 12p between products; a component of an Fq2 product is a SUM of two
// prime-field products, which would double every bound against the 169 p^2 a Montgomery product accepts.  Instead every addition and
// subtraction here reduces its result (one carry pass, one conditional subtraction of 2p: ~60 cheap VALU operations per component against
// the 486 multiply-adds of an Fq2 product), so every value that reaches a multiplier is "exact" and below 2p:
//     c0 = a0 b0 + a1 (3p - b1)   <= 2p 2p + 2p 3p = 10 p^2        c1 = a0 b1 + a1 b0 <= 8 p^2        (fq_mul2 of the prime field: one
//     Montgomery reduction per component, 2 x 243 multiply-adds);   squares: c0 = (a0 + a1)(a0 - a1 + 3p) <= 20 p^2, c1 = (2 a0) a1 <= 8 p^2;
//     fq_mul2 (a b + c d): four products per component under one reduction (fq_mul4 of the prime field), <= 20 p^2.
// The K of fq_sub<K> (how many p the prime-field version adds) is accepted and ignored; fq_norm and fq_tidy are the identity.
#ifndef MSM_BASE_NS
#error "fq2.h: MSM_BASE_NS (the namespace of the prime-field unit) is not defined"
#endif
#include <cstdint>

namespace MSM_FIELD_NS {

namespace fpn = MSM_BASE_NS;
using fp = fpn::fq;
constexpr int FP_L = fpn::FQ_L;          // limbs of a component
constexpr int FP_WORDS = fpn::FQ_WORDS;  // packed words of a component
static_assert(FQ_LIMBS == 2 * FP_L && FQ_WORDS == 2 * FP_WORDS && FQ_LIMB_BITS == fpn::FQ_W, "Fq2 layout");

struct fq {
  uint32_t v[2 * FP_L];
};
constexpr int FQ_L = FQ_LIMBS, FQ_W = FQ_LIMB_BITS;

FQ_HD fp f2_c0(const fq& a) {
  fp r;
#pragma unroll
  for (int i = 0; i < FP_L; i++) r.v[i] = a.v[i];
  return r;
}
FQ_HD fp f2_c1(const fq& a) {
  fp r;
#pragma unroll
  for (int i = 0; i < FP_L; i++) r.v[i] = a.v[FP_L + i];
  return r;
}
FQ_HD fq f2_make(const fp& c0, const fp& c1) {
  fq r;
#pragma unroll
  for (int i = 0; i < FP_L; i++) {
    r.v[i] = c0.v[i];
    r.v[FP_L + i] = c1.v[i];
  }
  return r;
}

// ---- the prime field's reduced addition / subtraction: operands exact (limbs < 2^29) and below 2p, result exact and below 2p
// s exact, value < 4p  ->  s or s - 2p, below 2p
FQ_HD fp fp_cond_sub_2p(const fp& s) {
  fp d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < FP_L; i++) {
    const uint32_t t = s.v[i] - FQ2_2P29[i] - borrow;
    borrow = t >> 31;
    d.v[i] = (i < FP_L - 1) ? (t & FQ_MASK) : t;
  }
  fp r;
#pragma unroll
  for (int i = 0; i < FP_L; i++) r.v[i] = borrow ? s.v[i] : d.v[i];
  return r;
}
FQ_HD fp fp_add_red(const fp& a, const fp& b) {
  fp s;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < FP_L; i++) {
    const uint32_t t = a.v[i] + b.v[i] + carry;
    s.v[i] = (i < FP_L - 1) ? (t & FQ_MASK) : t;
    carry = (i < FP_L - 1) ? (t >> FQ_W) : 0u;
  }
  return fp_cond_sub_2p(s);
}
FQ_HD fp fp_sub_red(const fp& a, const fp& b) {  // a - b + 2p in (0, 4p), then as above
  fp s;
  int32_t carry = 0;
#pragma unroll
  for (int i = 0; i < FP_L; i++) {
    const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)FQ2_2P29[i] + carry;
    s.v[i] = (i < FP_L - 1) ? ((uint32_t)t & FQ_MASK) : (uint32_t)t;
    carry = (i < FP_L - 1) ? (t >> FQ_W) : 0;  // arithmetic shift: -1, 0 or 1
  }
  FQ_ASSERT((s.v[FP_L - 1] >> 31) == 0, "fp_sub_red: negative value (subtrahend >= 2p)");
  return fp_cond_sub_2p(s);
}

// ---- the field interface of g1.h / msm_kernels.h over Fq2
FQ_HD fq fq_zero() {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = 0;
  return r;
}
FQ_HD fq fq_one() {  // Montgomery form of 1 + 0 u
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = FQ_ONE29[i];
  return r;
}
FQ_HD fq fq_norm(const fq& x) { return x; }
FQ_HD fq fq_tidy(const fq& x) { return x; }
FQ_HD fq fq_add(const fq& a, const fq& b) { return f2_make(fp_add_red(f2_c0(a), f2_c0(b)), fp_add_red(f2_c1(a), f2_c1(b))); }
FQ_HD fq fq_dbl(const fq& a) { return fq_add(a, a); }
template <int K>
FQ_HD fq fq_sub(const fq& a, const fq& b) {
  return f2_make(fp_sub_red(f2_c0(a), f2_c0(b)), fp_sub_red(f2_c1(a), f2_c1(b)));
}

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u : one two-product Montgomery reduction per component
#define FQ2_DEFINE_MUL(NAME, MUL2)                                            \
  FQ_HD fq NAME(const fq& a, const fq& b) {                                   \
    const fp a0 = f2_c0(a), a1 = f2_c1(a), b0 = f2_c0(b), b1 = f2_c1(b);      \
    const fp nb1 = fpn::fq_sub<3>(fpn::fq_zero(), b1); /* 3p - b1, normal */  \
    return f2_make(fpn::MUL2(a0, b0, a1, nb1), fpn::MUL2(a0, b1, a1, b0));    \
  }
FQ2_DEFINE_MUL(fq_mul, fq_mul2)
FQ2_DEFINE_MUL(fq_mul_fast, fq_mul2_fast)
#undef FQ2_DEFINE_MUL
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u
#define FQ2_DEFINE_SQR(NAME, MUL)                                                                     \
  FQ_HD fq NAME(const fq& a) {                                                                        \
    const fp a0 = f2_c0(a), a1 = f2_c1(a);                                                            \
    return f2_make(fpn::MUL(fpn::fq_add(a0, a1), fpn::fq_sub<3>(a0, a1)), fpn::MUL(fpn::fq_dbl(a0), a1)); \
  }
FQ2_DEFINE_SQR(fq_sqr, fq_mul)
FQ2_DEFINE_SQR(fq_sqr_fast, fq_mul_fast)
#undef FQ2_DEFINE_SQR
// a b + c d over Fq2: every component is FOUR prime-field products under one Montgomery reduction (round 5: 2 x 5 L^2 = 810 multiply-adds with 9
// limbs instead of the 972 of two Fq2 products, and no addition pass):
//     c0 = a0 b0 + a1 (3p - b1) + c0 d0 + c1 (3p - d1) <= 20 p^2        c1 = a0 b1 + a1 b0 + c0 d1 + c1 d0 <= 16 p^2
#define FQ2_DEFINE_MUL2(NAME, MUL4)                                                                     \
  FQ_HD fq NAME(const fq& a, const fq& b, const fq& c, const fq& d) {                                   \
    const fp a0 = f2_c0(a), a1 = f2_c1(a), b0 = f2_c0(b), b1 = f2_c1(b);                                \
    const fp c0 = f2_c0(c), c1 = f2_c1(c), d0 = f2_c0(d), d1 = f2_c1(d);                                \
    const fp nb1 = fpn::fq_sub<3>(fpn::fq_zero(), b1), nd1 = fpn::fq_sub<3>(fpn::fq_zero(), d1);        \
    return f2_make(fpn::MUL4(a0, b0, a1, nb1, c0, d0, c1, nd1), fpn::MUL4(a0, b1, a1, b0, c0, d1, c1, d0)); \
  }
FQ2_DEFINE_MUL2(fq_mul2, fq_mul4)
FQ2_DEFINE_MUL2(fq_mul2_fast, fq_mul4_fast)
#undef FQ2_DEFINE_MUL2
FQ_HD void fq_mul_fast_ip(fq& a, const fq& b) { a = fq_mul_fast(a, b); }
FQ_HD void fq_mul2_fast_ip(const fq& a, const fq& b, fq& c, const fq& d) { c = fq_mul2_fast(a, b, c, d); }

FQ_HD bool fq_is_zero_exact(const fq& x) { return fpn::fq_is_zero_exact(f2_c0(x)) && fpn::fq_is_zero_exact(f2_c1(x)); }
FQ_HD fq fq_canonical(const fq& x) { return f2_make(fpn::fq_canonical(f2_c0(x)), fpn::fq_canonical(f2_c1(x))); }
FQ_HD fq fq_neg_canonical(const fq& y) { return f2_make(fpn::fq_neg_canonical(f2_c0(y)), fpn::fq_neg_canonical(f2_c1(y))); }
// (the prime-field version returns 2p - y with lazy limbs for a multiplier operand; here operands stay exact: -y canonical)
FQ_HD fq fq_neg_lazy(const fq& y) { return fq_neg_canonical(y); }

FQ_HD fq fq_unpack(const uint32_t w[FQ_WORDS]) { return f2_make(fpn::fq_unpack(w), fpn::fq_unpack(w + FP_WORDS)); }
FQ_HD void fq_pack(uint32_t w[FQ_WORDS], const fq& x) {
  fpn::fq_pack(w, f2_c0(x));
  fpn::fq_pack(w + FP_WORDS, f2_c1(x));
}
FQ_HD fq fq_to_mont(const fq& x) { return f2_make(fpn::fq_to_mont(f2_c0(x)), fpn::fq_to_mont(f2_c1(x))); }
FQ_HD fq fq_from_mont256(const fq& x) { return f2_make(fpn::fq_from_mont256(f2_c0(x)), fpn::fq_from_mont256(f2_c1(x))); }
FQ_HD fq fq_from_mont(const fq& x) { return f2_make(fpn::fq_from_mont(f2_c0(x)), fpn::fq_from_mont(f2_c1(x))); }

}  // namespace MSM_FIELD_NS
