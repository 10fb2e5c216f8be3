// BN254 base-field arithmetic for gfx950: 9 limbs x 29 bits in 32-bit VGPRs, Montgomery radix R = 2^261.
//
// Why this shape (measured on MI355X, profiles/r01_ubench_valu_rates.txt): v_mad_u64_u32 issues at ~2.3 ns per
// wave-instruction per SIMD -- only 1.25x a plain VOP3 op -- while every carry-propagating add
// (v_add_co_u32 / v_addc_co_u32) costs ~2.0 ns, i.e. nearly a full multiply.  A saturated 8 x 32-bit limb
// multiplier therefore spends as much on carries as on products.  With 29-bit limbs a 64-bit column
// accumulator absorbs all 18 partial products of a Montgomery step without any carry instruction
// (18 * 2^58 < 2^64), additions are 9 independent VOP2 adds with no carry chain ("lazy" limbs up to 2^30),
// and 2^261 / p = 169 leaves room for values up to ~12p between multiplications, so nothing is reduced
// mod p until a value is stored.
//
// The reference's field library (src/cuzk/wgsl/field/field.template.wgsl:40-135,
// src/cuzk/wgsl/montgomery/mont_pro_product.template.wgsl:7-53, src/cuzk/wgsl/bigint/bigint.template.wgsl:11-46)
// uses 20 x 13-bit limbs because WGSL has no 64-bit integers; only its semantics (Montgomery product,
// add, sub mod p) are kept here.
//
// The limb layout is a parameter of the curve unit (FQ_LIMBS limbs of FQ_LIMB_BITS bits, FQ_WORDS packed 32-bit words per element,
// from the generated constants): 9 x 29 bits for the 254 / 255-bit fields, 14 x 28 bits for BLS12-381's 381-bit field (28 x 2^58 partial
// products and lazy squares must still fit a 64-bit column: 14 limbs of 29 bits do not).  The comments below quote the 9 x 29 numbers.
//
// Conventions
//   value(x)  = sum_i x.v[i] * 2^(29 i)
//   "normal"  : limbs 0..7 < 2^29 + 8, top limb small       (output of fq_sub / fq_norm)
//   "exact"   : limbs 0..7 < 2^29 exactly (unique)          (output of fq_mul / fq_sqr; value < 2p)
//   "lazy"    : limbs < 2^30 + 16                           (fq_add of two normal values)
//   fq_mul / fq_sqr accept lazy operands with value(a) * value(b) <= 169 p^2 and return exact, < 2p.
#ifndef MSM_CURVE_UNIT  // stand-alone use instantiates BN254, once; inside a curve unit (curve_unit.h) the unit includes the parts
#pragma once
#include "curve_select.h"
#include MSM_CURVE_CONSTANTS
#endif
#include <cstdint>

#if defined(__HIPCC__)
#define FQ_HD __host__ __device__ __forceinline__
#else
#define FQ_HD inline
#endif

#if defined(FQ_CHECK)  // host-only bound checking used by tests/test_fq29_host.py
#include <cstdio>
#include <cstdlib>
#define FQ_ASSERT(c, msg)                                      \
  do {                                                         \
    if (!(c)) {                                                \
      fprintf(stderr, "fq29 bound violated: %s\n", msg);       \
      abort();                                                 \
    }                                                          \
  } while (0)
#else
#define FQ_ASSERT(c, msg) ((void)0)
#endif

namespace MSM_FIELD_NS {

struct fq {
  uint32_t v[FQ_LIMBS];
};
constexpr int FQ_L = FQ_LIMBS, FQ_W = FQ_LIMB_BITS;  // short names for the loops below

}  // namespace MSM_FIELD_NS
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FQ29_NO_ASM)
#define FQ29_ASM 1
#ifndef MSM_FQ_ASM_HEADER
#define MSM_FQ_ASM_HEADER "fq29_asm.h"  // 9 x 29-bit limbs; a unit with another limb layout names its own generated file
#endif
#include MSM_FQ_ASM_HEADER  // fq_mul_asm / fq_sqr_asm: the multipliers as single inline-assembly blocks (tools/gen_fq29_asm.py)
#endif
namespace MSM_FIELD_NS {

FQ_HD fq fq_zero() {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = 0;
  return r;
}
FQ_HD fq fq_one() {  // Montgomery form of 1
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = FQ_ONE29[i];
  return r;
}

// One carry-save pass: limbs 0..7 end < 2^29 + (max_in >> 29); value unchanged.
FQ_HD fq fq_norm(const fq& x) {
  fq r;
  r.v[0] = x.v[0] & FQ_MASK;
#pragma unroll
  for (int i = 1; i < FQ_L - 1; i++) r.v[i] = (x.v[i] & FQ_MASK) + (x.v[i - 1] >> FQ_W);
  r.v[FQ_L - 1] = x.v[FQ_L - 1] + (x.v[FQ_L - 2] >> FQ_W);
  return r;
}

// Lazy add: no carries, no reduction.
FQ_HD fq fq_add(const fq& a, const fq& b) {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
FQ_HD fq fq_dbl(const fq& a) { return fq_add(a, a); }

// a - b + K*p, normal.  Requires value(b) < (K-1)*p and b's limbs 0..7 <= 2^31 - 4, a's <= 2^30 + 2^29.
template <int K>
FQ_HD fq fq_sub(const fq& a, const fq& b) {
  static_assert(K >= 2 && K <= 16, "K*p constant not tabulated");
  fq t;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    FQ_ASSERT(FQ_KP29[K][i] >= b.v[i] || i == FQ_L - 1, "fq_sub: limb borrow");
    FQ_ASSERT(i != FQ_L - 1 || FQ_KP29[K][FQ_L - 1] + a.v[FQ_L - 1] >= b.v[FQ_L - 1], "fq_sub: top limb borrow");
    t.v[i] = a.v[i] + FQ_KP29[K][i] - b.v[i];
  }
  return fq_norm(t);
}

#if defined(FQ_CHECK)
// host-only (tests/test_fq29_host.py): r has exact limbs; is its value below 2p?  A Montgomery product's result is < 2p exactly when
// its operands respected value(a) * value(b) <= (2^261 / p) p^2 -- the contract every bound comment in g1.h relies on.
inline bool fq_check_below_2p(const fq& r) {
  uint32_t two_p[FQ_L];
  uint64_t carry = 0;
  for (int i = 0; i < FQ_L; i++) {
    const uint64_t t = 2ull * FQ_P29[i] + carry;
    two_p[i] = (uint32_t)(i < FQ_L - 1 ? t & FQ_MASK : t);
    carry = i < FQ_L - 1 ? t >> FQ_W : 0;
  }
  for (int i = FQ_L - 1; i >= 0; i--) {
    if (r.v[i] != two_p[i]) return r.v[i] < two_p[i];
  }
  return false;
}
#endif

// Montgomery product a*b/R mod p, R = 2^261.  Operand limbs <= 2^30 + 16; value(a)*value(b) <= 169 p^2.
// Result exact (limbs < 2^29), value < 2p.
FQ_HD fq fq_mul(const fq& a, const fq& b) {
#if defined(FQ29_ASM) && !defined(FQ29_ASM_SMVP_ONLY)  // the assembly multipliers in every kernel (round 4; -DFQ29_ASM_SMVP_ONLY: only in the SMVP's mixed addition, as rounds 1 - 3)
  return fq_mul_asm(a, b);
#endif
  uint64_t c[2 * FQ_L];
#pragma unroll
  for (int k = 0; k < 2 * FQ_L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    FQ_ASSERT(a.v[i] <= (1u << (FQ_W + 1)) + 64 && b.v[i] <= (1u << (FQ_W + 1)) + 64, "fq_mul: operand limb too large");
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)a.v[j] * b.v[i];
    const uint32_t m = ((uint32_t)c[i] * FQ_N0_29) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)m * FQ_P29[j];
    c[i + 1] += c[i] >> FQ_W;
  }
  fq r;
#pragma unroll
  for (int k = FQ_L; k < 2 * FQ_L - 1; k++) {
    r.v[k - FQ_L] = (uint32_t)c[k] & FQ_MASK;
    c[k + 1] += c[k] >> FQ_W;
  }
  r.v[FQ_L - 1] = (uint32_t)c[2 * FQ_L - 1];
  FQ_ASSERT((c[2 * FQ_L - 1] >> (FQ_W - 3)) == 0, "fq_mul: result >= 2^258");
  FQ_ASSERT(fq_check_below_2p(r), "fq_mul: result >= 2p (operand value bound violated)");
  return r;
}

// (a*b + c*d)/R mod p with ONE Montgomery reduction (243 multiply-adds instead of 324).  b, c, d must be normal or exact
// (limbs < 2^29 + 8), a may be lazy (limbs < 2^30 + 16): a column then holds at most 9 products < 2^59.01, 9 products
// < 2^58.01 and 9 reduction terms < 2^58, together < 2^63.2.  value(a)*value(b) + value(c)*value(d) <= 169 p^2.
// Result exact, < 2p.
FQ_HD fq fq_mul2(const fq& a, const fq& b, const fq& c_, const fq& d) {
#if defined(FQ29_ASM) && !defined(FQ29_ASM_SMVP_ONLY)
  return fq_mul2_asm(a, b, c_, d);
#endif
  uint64_t c[2 * FQ_L];
#pragma unroll
  for (int k = 0; k < 2 * FQ_L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    FQ_ASSERT(a.v[i] <= (1u << (FQ_W + 1)) + 64 && b.v[i] <= (1u << FQ_W) + 64 && c_.v[i] <= (1u << FQ_W) + 64 && d.v[i] <= (1u << FQ_W) + 64,
              "fq_mul2: operand limb too large");
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)a.v[j] * b.v[i];
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)c_.v[j] * d.v[i];
    const uint32_t m = ((uint32_t)c[i] * FQ_N0_29) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)m * FQ_P29[j];
    c[i + 1] += c[i] >> FQ_W;
  }
  fq r;
#pragma unroll
  for (int k = FQ_L; k < 2 * FQ_L - 1; k++) {
    r.v[k - FQ_L] = (uint32_t)c[k] & FQ_MASK;
    c[k + 1] += c[k] >> FQ_W;
  }
  r.v[FQ_L - 1] = (uint32_t)c[2 * FQ_L - 1];
  FQ_ASSERT((c[2 * FQ_L - 1] >> (FQ_W - 3)) == 0, "fq_mul2: result >= 2^258");
  FQ_ASSERT(fq_check_below_2p(r), "fq_mul2: result >= 2p (operand value bound violated)");
  return r;
}

// (a*b + c*d + e*f + g*h)/R mod p with ONE Montgomery reduction -- a component of the Fq2 form of fq_mul2 (csrc/fq2.h: Y3 = R T + (-Y) PPP over
// Fq2 is four prime-field products per component; round 5: 5 L^2 multiply-adds instead of the 6 L^2 of two fq_mul2).  EVERY operand exact (limbs
// < 2^W + 64): a column then holds at most 4 L products and L reduction terms below 2^(2W), within 64 bits for both limb layouts (the generator
// asserts it).  Sum of the four products <= R / p * p^2 (169 p^2 for BN254; the Fq2 formulas stay below 24 p^2).  Result exact, < 2p.
FQ_HD fq fq_mul4(const fq& a, const fq& b, const fq& c_, const fq& d, const fq& e, const fq& f, const fq& g, const fq& h) {
#if defined(FQ29_ASM) && !defined(FQ29_ASM_SMVP_ONLY)
  return fq_mul4_asm(a, b, c_, d, e, f, g, h);
#endif
  uint64_t c[2 * FQ_L];
#pragma unroll
  for (int k = 0; k < 2 * FQ_L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    FQ_ASSERT(a.v[i] <= (1u << FQ_W) + 64 && b.v[i] <= (1u << FQ_W) + 64 && c_.v[i] <= (1u << FQ_W) + 64 && d.v[i] <= (1u << FQ_W) + 64 &&
                  e.v[i] <= (1u << FQ_W) + 64 && f.v[i] <= (1u << FQ_W) + 64 && g.v[i] <= (1u << FQ_W) + 64 && h.v[i] <= (1u << FQ_W) + 64,
              "fq_mul4: operand limb too large");
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)a.v[j] * b.v[i];
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)c_.v[j] * d.v[i];
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)e.v[j] * f.v[i];
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)g.v[j] * h.v[i];
    const uint32_t m = ((uint32_t)c[i] * FQ_N0_29) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)m * FQ_P29[j];
    c[i + 1] += c[i] >> FQ_W;
  }
  fq r;
#pragma unroll
  for (int k = FQ_L; k < 2 * FQ_L - 1; k++) {
    r.v[k - FQ_L] = (uint32_t)c[k] & FQ_MASK;
    c[k + 1] += c[k] >> FQ_W;
  }
  r.v[FQ_L - 1] = (uint32_t)c[2 * FQ_L - 1];
  FQ_ASSERT(fq_check_below_2p(r), "fq_mul4: result >= 2p (operand value bound violated)");
  return r;
}
FQ_HD fq fq_mul4_fast(const fq& a, const fq& b, const fq& c_, const fq& d, const fq& e, const fq& f, const fq& g, const fq& h) {
#if defined(FQ29_ASM)
  return fq_mul4_asm(a, b, c_, d, e, f, g, h);
#else
  return fq_mul4(a, b, c_, d, e, f, g, h);
#endif
}

// Montgomery square: 45 products instead of 81 (cross terms doubled once).
FQ_HD fq fq_sqr(const fq& a) {
#if defined(FQ29_ASM) && !defined(FQ29_ASM_SMVP_ONLY)
  return fq_sqr_asm(a);
#endif
  uint64_t c[2 * FQ_L];
  uint32_t a2[FQ_L];
#pragma unroll
  for (int k = 0; k < 2 * FQ_L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    FQ_ASSERT(a.v[i] <= (1u << (FQ_W + 1)) + 64, "fq_sqr: operand limb too large");
    a2[i] = a.v[i] << 1;
  }
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    c[2 * i] += (uint64_t)a.v[i] * a.v[i];
#pragma unroll
    for (int j = i + 1; j < FQ_L; j++) c[i + j] += (uint64_t)a.v[i] * a2[j];
  }
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    const uint32_t m = ((uint32_t)c[i] * FQ_N0_29) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_L; j++) c[i + j] += (uint64_t)m * FQ_P29[j];
    c[i + 1] += c[i] >> FQ_W;
  }
  fq r;
#pragma unroll
  for (int k = FQ_L; k < 2 * FQ_L - 1; k++) {
    r.v[k - FQ_L] = (uint32_t)c[k] & FQ_MASK;
    c[k + 1] += c[k] >> FQ_W;
  }
  r.v[FQ_L - 1] = (uint32_t)c[2 * FQ_L - 1];
  FQ_ASSERT(fq_check_below_2p(r), "fq_sqr: result >= 2p (operand value bound violated)");
  return r;
}

// The same products for the hottest loop (the SMVP's mixed addition): on the device one inline-assembly block each, in which
// every column's multiply-add chain starts from the previous carry (17 fewer VALU instructions than the compiler's form).
FQ_HD fq fq_mul_fast(const fq& a, const fq& b) {
#if defined(FQ29_ASM)
  return fq_mul_asm(a, b);
#else
  return fq_mul(a, b);
#endif
}
FQ_HD fq fq_mul2_fast(const fq& a, const fq& b, const fq& c_, const fq& d) {
#if defined(FQ29_ASM)
  return fq_mul2_asm(a, b, c_, d);
#else
  return fq_mul2(a, b, c_, d);
#endif
}
FQ_HD fq fq_sqr_fast(const fq& a) {
#if defined(FQ29_ASM)
  return fq_sqr_asm(a);
#else
  return fq_sqr(a);
#endif
}

// ... and with the result written over one operand (a = a b ; c = a b + c d): on the device the assembly block produces result limb j in
// the register of the operand's limb j, which is dead by then -- a loop-carried accumulator is updated without a copy back
FQ_HD void fq_mul_fast_ip(fq& a, const fq& b) {
#if defined(FQ29_ASM)
  fq_mul_ip_asm(a, b);
#else
  a = fq_mul(a, b);
#endif
}
FQ_HD void fq_mul2_fast_ip(const fq& a, const fq& b, fq& c_, const fq& d) {
#if defined(FQ29_ASM)
  fq_mul2_ip_asm(a, b, c_, d);
#else
  c_ = fq_mul2(a, b, c_, d);
#endif
}

// x exact and < 2p  ->  true iff x == 0 (mod p)
FQ_HD bool fq_is_zero_exact(const fq& x) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    z |= x.v[i];
    e |= x.v[i] ^ FQ_P29[i];
  }
  return z == 0 || e == 0;
}

// x exact and < 2p  ->  canonical representative in [0, p), exact
FQ_HD fq fq_canonical(const fq& x) {
  fq d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    const uint32_t t = x.v[i] - FQ_P29[i] - borrow;
    borrow = t >> 31;
    d.v[i] = (i < FQ_L - 1) ? (t & FQ_MASK) : t;
  }
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = borrow ? x.v[i] : d.v[i];
  return r;
}

// any normal/lazy x with value <= 84p (so that x * 2p <= 169 p^2)  ->  exact value mod p in [0, 2p), same Montgomery form
FQ_HD fq fq_tidy(const fq& x) { return fq_mul(x, fq_one()); }

// 8 x 32-bit packed words (value < 2^256)  <->  9 x 29-bit limbs (exact)
FQ_HD fq fq_unpack(const uint32_t w[FQ_WORDS]) {
  fq r;
  uint64_t acc = 0;
  int bits = 0, k = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    if (bits < FQ_W && k < FQ_WORDS) {
      acc |= (uint64_t)w[k++] << bits;
      bits += 32;
    }
    r.v[i] = (uint32_t)acc & FQ_MASK;
    acc >>= FQ_W;
    bits -= FQ_W;
  }
  return r;
}
FQ_HD void fq_pack(uint32_t w[FQ_WORDS], const fq& x) {  // x exact, value < 2^(32 FQ_WORDS)
  uint64_t acc = 0;
  int bits = 0, k = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    acc |= (uint64_t)x.v[i] << bits;
    bits += FQ_W;
    if (bits >= 32 && k < FQ_WORDS) {
      w[k++] = (uint32_t)acc;
      acc >>= 32;
      bits -= 32;
    }
  }
}

// canonical integer (packed) -> Montgomery form, and back
FQ_HD fq fq_to_mont(const fq& x_plain) {
  fq r2;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r2.v[i] = FQ_R2_29[i];
  return fq_canonical(fq_mul(x_plain, r2));
}
// x * 2^256 mod p (the in-memory form of a 4 x 64-bit Montgomery library with R = 2^256, canonical) -> device Montgomery form
FQ_HD fq fq_from_mont256(const fq& x_r256) {
  fq c;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) c.v[i] = FQ_2P266_29[i];
  return fq_canonical(fq_mul(x_r256, c));  // x 2^256 * 2^266 / 2^261 = x 2^261
}
FQ_HD fq fq_from_mont(const fq& x) {  // x normal, value <= 84p
  fq one = fq_zero();
  one.v[0] = 1;
  return fq_canonical(fq_mul(x, one));
}

// y exact (limbs < 2^29), value <= p  ->  2p - y with lazy limbs (< 2^30), no borrow chain: fit for a multiplier operand
FQ_HD fq fq_neg_lazy(const fq& y) {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    FQ_ASSERT(FQ_2P_LAZY29[i] >= y.v[i], "fq_neg_lazy: limb borrow");
    r.v[i] = FQ_2P_LAZY29[i] - y.v[i];
  }
  return r;
}

FQ_HD fq fq_neg_canonical(const fq& y) {  // y canonical in [0,p) -> p - y (or 0)
  uint32_t z = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) z |= y.v[i];
  fq t;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    const uint32_t d = FQ_P29[i] - y.v[i] - borrow;
    borrow = d >> 31;
    t.v[i] = (i < FQ_L - 1) ? (d & FQ_MASK) : d;
  }
  return z ? t : y;
}

}  // namespace MSM_FIELD_NS
