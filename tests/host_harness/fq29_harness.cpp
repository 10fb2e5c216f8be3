// Host build (g++ -DFQ_CHECK) of the device arithmetic headers, for CPU-side unit tests of the exact code
// the HIP kernels inline.  Test-only: not part of libmsm_hip.so.
#include <cstring>

#ifndef HARNESS_PRELUDE_DONE  // (fq2_harness.cpp instantiates the G2 unit's headers itself, then includes this file)
#include "g1.h"
#include "glv.h"
#endif

#ifndef HARNESS_FIELD_NS  // -DMSM_FIELD_NS=... -DMSM_CURVE_CONSTANTS=... -DHARNESS_FIELD_NS=...: the same harness for another curve
#define HARNESS_FIELD_NS bn254
#endif
using namespace HARNESS_FIELD_NS;
constexpr int HB = 4 * FQ_WORDS;  // bytes of a field element on the wire (32; 48 for BLS12-381); points 2 HB, Jacobian records 3 HB

static fq load_fq(const uint8_t* b) {  // canonical LE bytes -> Montgomery fq
  uint32_t w[FQ_WORDS];
  memcpy(w, b, HB);
  return fq_to_mont(fq_unpack(w));
}
static void store_fq(uint8_t* b, const fq& x) {  // normal Montgomery fq (value <= 84p) -> canonical LE bytes
  uint32_t w[FQ_WORDS];
  fq_pack(w, fq_from_mont(x));
  memcpy(b, w, HB);
}
static g1_xyzz load_jac(const uint8_t* b) { return g1_from_jacobian(load_fq(b), load_fq(b + HB), load_fq(b + 2 * HB)); }
static void store_jac(uint8_t* b, const g1_xyzz& p) {
  fq X, Y, Z;
  g1_to_jacobian(p, X, Y, Z);
  store_fq(b, X);
  store_fq(b + HB, Y);
  store_fq(b + 2 * HB, Z);
}

extern "C" {
#ifndef MSM_FQ2
// scalar split of the curve endomorphism (csrc/glv.h): n x 32 B scalars -> n x (16 B half 1 | 16 B half 2), sign in bit 127;
// returns the number of scalars whose halves did not fit
size_t h_glv_split(const uint8_t* scalars, uint8_t* out, size_t n) {
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t k[8], h1[4], h2[4];
    memcpy(k, scalars + 32 * i, 32);
    if (!glv_split(k, h1, h2)) bad++;
    memcpy(out + 32 * i, h1, 16);
    memcpy(out + 32 * i + 16, h2, 16);
  }
  return bad;
}
// x -> beta x (the endomorphism's action on a point's x coordinate), canonical bytes
void h_fq_mul_beta(const uint8_t* a, uint8_t* out, size_t n) {
  fq beta;
  for (int i = 0; i < FQ_LIMBS; i++) beta.v[i] = FQ_BETA29[i];
  for (size_t i = 0; i < n; i++) store_fq(out + HB * i, fq_mul(load_fq(a + HB * i), beta));
}
#endif
// op: 0 add, 1 sub, 2 mul, 3 sqr, 4 neg
void h_fq_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    fq x = load_fq(a + HB * i), y = b ? load_fq(b + HB * i) : fq_zero(), z;
    switch (op) {
      case 0: z = fq_add(x, y); break;
      case 1: z = fq_sub<2>(x, y); break;
      case 2: z = fq_mul(x, y); break;
      case 3: z = fq_sqr(x); break;
      default: z = fq_neg_canonical(x); break;
    }
    store_fq(out + HB * i, z);
  }
}
// pack/unpack round trip of raw 256-bit values
void h_fq_roundtrip(const uint8_t* a, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint32_t w[FQ_WORDS], w2[FQ_WORDS];
    memcpy(w, a + HB * i, HB);
    fq_pack(w2, fq_unpack(w));
    memcpy(out + HB * i, w2, HB);
  }
}
// acc (96 B Jacobian canonical) += chain of `m` affine points (64 B each); exercises the lazy bounds over long chains
void h_g1_madd_chain(const uint8_t* acc, const uint8_t* pts, size_t m, uint8_t* out) {
  g1_xyzz a = load_jac(acc);
  for (size_t i = 0; i < m; i++) g1_madd(a, load_fq(pts + 2 * HB * i), load_fq(pts + 2 * HB * i + HB));
  store_jac(out, a);
}
// the SMVP's signed-state form: acc += (negs[i] ? -pts[i] : pts[i]) through g1_madd_w, sign applied at the end
void h_g1_madd_w_chain(const uint8_t* acc, const uint8_t* pts, const uint8_t* negs, size_t m, uint8_t* out) {
  g1_xyzz a = load_jac(acc);
  bool wneg = false;
  for (size_t i = 0; i < m; i++) g1_madd_w(a, wneg, load_fq(pts + 2 * HB * i), load_fq(pts + 2 * HB * i + HB), negs[i] != 0);
  store_jac(out, g1_unsigned(a, wneg));
}
#ifndef MSM_FQ2  // (csrc/fq2.h reduces after every addition: there are no lazy bounds to push)
// The value bounds g1.h promises between operations (X < 9p, Y < 5p, ZZ < 2p, ZZZ < 2p), pushed to their limits: the accumulator's
// coordinates are raised by kx / ky / kz multiples of p (same residues, larger representatives) before each operation; every Montgomery
// result is asserted below 2p (-DFQ_CHECK), i.e. the operand bounds of every multiplication in the formulas hold at the edge.
static fq raise_by_p(fq x, int k) {
  fq pp;
  for (int i = 0; i < FQ_LIMBS; i++) pp.v[i] = FQ_P29[i];
  for (int j = 0; j < k; j++) x = fq_norm(fq_add(x, pp));
  return x;
}
// op 0: g1_madd_w (sign state `wneg` on entry), 1: g1_madd, 2: g1_add (b raised too), 3: g1_double
void h_g1_at_the_bounds(int op, const uint8_t* acc, const uint8_t* other, int neg, int wneg_in, int kx, int ky, int kz, uint8_t* out) {
  g1_xyzz a = load_jac(acc);
  if (!a.inf) {
    a.x = raise_by_p(fq_canonical(a.x), kx);
    a.y = raise_by_p(fq_canonical(a.y), ky);
    a.zz = raise_by_p(fq_canonical(a.zz), kz);    // kz <= 1: exact limbs, < 2p
    a.zzz = raise_by_p(fq_canonical(a.zzz), kz);
  }
  if (op == 0) {
    bool wneg = wneg_in != 0;  // the accumulator then stands for (X, -Y): the caller passes the point it means negated
    g1_madd_w(a, wneg, load_fq(other), load_fq(other + HB), neg != 0);
    store_jac(out, g1_unsigned(a, wneg));
  } else if (op == 1) {
    g1_madd(a, load_fq(other), load_fq(other + HB));
    store_jac(out, a);
  } else if (op == 2) {
    g1_xyzz b = load_jac(other);
    if (!b.inf) {
      b.x = raise_by_p(fq_canonical(b.x), kx);
      b.y = raise_by_p(fq_canonical(b.y), ky);
      b.zz = raise_by_p(fq_canonical(b.zz), kz);
      b.zzz = raise_by_p(fq_canonical(b.zzz), kz);
    }
    store_jac(out, g1_add(a, b));
  } else {
    store_jac(out, g1_double(a));
  }
}
#endif
// op: 0 add, 1 double(a)
void h_g1_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    g1_xyzz p = load_jac(a + 3 * HB * i);
    g1_xyzz r = op == 0 ? g1_add(p, load_jac(b + 3 * HB * i)) : g1_double(p);
    store_jac(out + 3 * HB * i, r);
  }
}
void h_g1_mul_u32(const uint8_t* a, uint32_t k, uint8_t* out) { store_jac(out, g1_mul_u32(load_jac(a), k)); }
// running-sum chain: feeds outputs of g1_add back into g1_add many times (bound stability)
void h_g1_running_sum(const uint8_t* pts96, size_t n, uint8_t* out) {
  g1_xyzz m = g1_identity(), g = g1_identity();
  for (size_t i = 0; i < n; i++) {
    m = g1_add(m, load_jac(pts96 + 3 * HB * i));
    g = g1_add(g, m);
  }
  store_jac(out, g);
}
}
