/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See oracle/bn254.h for scope, parity status and citations. */
#include "bn254.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------------
 * Fq : 4 x 64-bit Montgomery, R = 2^256  (the in-memory form halo2curves uses)
 * ---------------------------------------------------------------------------------------------- */
/* NL 64-bit limbs per base-field element (4; 6 for BLS12-381: bn254.h), CB bytes per coordinate on the wire, PB per affine point, JB per
 * Jacobian record; scalars are 32 bytes on every curve */
#define NL ONL
#define FB (8 * NL) /* bytes of a prime-field element */
#if defined(ORACLE_G2)
/* -DORACLE_G2 (with or without -DORACLE_BLS12_381): the same restatement over G2, the order-r subgroup of the sextic twist over
 * Fq2 = Fq[u] / (u^2 + 1) (SURVEY.md 8f-4 "other curves / G2"; halo2curves' msm_best is generic over CurveAffine, so the reference's CPU path
 * applied to G2Affine is this algorithm): the prime field below becomes `fp_*` / `ofp`, and `ofq` / `fq_*` -- what the group code, the MSM and
 * the stage models are written against -- are Fq2 on top of it.  A coordinate is c0 || c1 on the wire. */
#define CB (2 * FB)
#define BF(name) fp_##name
typedef ofp bfe;
#else
#define CB FB
#define BF(name) fq_##name
typedef ofq bfe;
#endif
#define PB (2 * CB)
#define JB (3 * CB)
#define FBITS (64 * NL)
#if defined(ORACLE_BLS12_381)
/* -DORACLE_BLS12_381: BLS12-381 G1 (reference README.md "Implement cuzk on other curves"): y^2 = x^3 + 4 over the 381-bit p
 * (6 x 64-bit limbs, R = 2^384, 48-byte coordinates), scalars modulo the 255-bit r; p = 3 mod 4. */
static const uint64_t FQ_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t FR_R[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t FQ_N0 = 0x89f3fffcfffcfffdull; /* -p^-1 mod 2^64 */
#define CURVE_B_IS_MINUS 0
#define CURVE_B_ABS 4
#elif !defined(ORACLE_GRUMPKIN) && !defined(ORACLE_PALLAS) && !defined(ORACLE_VESTA)
static const uint64_t FQ_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FR_R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FQ_N0 = 0x87d20782e4866389ull; /* -p^-1 mod 2^64 */
#define CURVE_B_IS_MINUS 0
#define CURVE_B_ABS 3 /* y^2 = x^3 + 3 */
#elif defined(ORACLE_GRUMPKIN)
/* -DORACLE_GRUMPKIN: the same restatement over Grumpkin, BN254's cycle partner (SURVEY.md 8f-4 "other curves"): base field =
 * BN254's scalar field r, scalar field = BN254's base field p, y^2 = x^3 - 17, generator (1, sqrt(-16)).  Only the moduli, the
 * curve constant and the sampler's square root (r = 1 mod 4: Tonelli-Shanks) differ. */
static const uint64_t FQ_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FR_R[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FQ_N0 = 0xc2e1f593efffffffull; /* -r^-1 mod 2^64 */
#define CURVE_B_IS_MINUS 1
#define CURVE_B_ABS 17 /* y^2 = x^3 - 17 */
#elif defined(ORACLE_PALLAS)
/* -DORACLE_PALLAS / -DORACLE_VESTA: the Pasta cycle (the reference keeps dead Pallas shaders, src/naive/wgsl/pallas, and lists other
 * curves as future work): 255-bit moduli p, q; Pallas: y^2 = x^3 + 5 over Fp with q points, Vesta: the same equation over Fq with p
 * points; generators (-1, 2).  2-adicity 32: Tonelli-Shanks in the sampler. */
static const uint64_t FQ_P[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull};
static const uint64_t FR_R[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull, 0x4000000000000000ull};
static const uint64_t FQ_N0 = 0x992d30ecffffffffull;
#define CURVE_B_IS_MINUS 0
#define CURVE_B_ABS 5
#else /* ORACLE_VESTA */
static const uint64_t FQ_P[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull, 0x4000000000000000ull};
static const uint64_t FR_R[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull};
static const uint64_t FQ_N0 = 0x8c46eb20ffffffffull;
#define CURVE_B_IS_MINUS 0
#define CURVE_B_ABS 5
#endif

static bfe BF_R1, BF_R2, BF_ZERO; /* prime field: R mod p, R^2 mod p, 0; filled by init */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static int ge_p(const uint64_t a[NL]) {
  for (int i = NL - 1; i >= 0; i--) {
    if (a[i] > FQ_P[i]) return 1;
    if (a[i] < FQ_P[i]) return 0;
  }
  return 1;
}
static void sub_p(uint64_t a[NL]) {
  u128 br = 0;
  for (int i = 0; i < NL; i++) {
    u128 t = (u128)a[i] - FQ_P[i] - br;
    a[i] = (uint64_t)t;
    br = (t >> 64) & 1;
  }
}
static void BF(add)(bfe* o, const bfe* a, const bfe* b) {
  u128 c = 0;
  uint64_t t[NL];
  for (int i = 0; i < NL; i++) {
    c += (u128)a->l[i] + b->l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  /* p < 2^254 so a + b < 2^255: no carry out */
  if (ge_p(t)) sub_p(t);
  memcpy(o->l, t, FB);
}
static void BF(sub)(bfe* o, const bfe* a, const bfe* b) {
  u128 br = 0;
  uint64_t t[NL];
  for (int i = 0; i < NL; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < NL; i++) {
      c += (u128)t[i] + FQ_P[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(o->l, t, FB);
}
static void BF(neg)(bfe* o, const bfe* a) { BF(sub)(o, &BF_ZERO, a); }
static void BF(dbl)(bfe* o, const bfe* a) { BF(add)(o, a, a); }
static int BF(is_zero)(const bfe* a) {
  uint64_t z = 0;
  for (int i = 0; i < NL; i++) z |= a->l[i];
  return z == 0;
}
static int BF(eq)(const bfe* a, const bfe* b) { return memcmp(a->l, b->l, FB) == 0; }

/* CIOS Montgomery product */
static void BF(mul)(bfe* o, const bfe* a, const bfe* b) {
  uint64_t t[NL + 2];
  memset(t, 0, sizeof t);
  for (int i = 0; i < NL; i++) {
    u128 c = 0;
    for (int j = 0; j < NL; j++) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[NL];
    t[NL] = (uint64_t)c;
    t[NL + 1] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * FQ_N0;
    c = (u128)m * FQ_P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < NL; j++) {
      c += (u128)m * FQ_P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[NL];
    t[NL - 1] = (uint64_t)c;
    t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
  }
  if (t[NL] || ge_p(t)) sub_p(t);
  memcpy(o->l, t, FB);
}
static void BF(sqr)(bfe* o, const bfe* a) { BF(mul)(o, a, a); }

static void BF(pow)(bfe* o, const bfe* a, const uint64_t e[NL]) {
  bfe acc = BF_R1, base = *a;
  for (int i = 0; i < FBITS; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) BF(mul)(&acc, &acc, &base);
    BF(sqr)(&base, &base);
  }
  *o = acc;
}
static void BF(inv)(bfe* o, const bfe* a) {
  uint64_t e[NL];
  memcpy(e, FQ_P, FB);
  e[0] -= 2;
  BF(pow)(o, a, e);
}
/* canonical LE bytes <-> Montgomery */
static int BF(from_bytes)(bfe* o, const uint8_t* b) {
  bfe t;
  memcpy(t.l, b, FB); /* little-endian host */
  int canonical = !ge_p(t.l);
  BF(mul)(o, &t, &BF_R2);
  return canonical;
}
static void BF(to_bytes)(uint8_t* b, const bfe* a) {
  bfe one, t;
  memset(&one, 0, sizeof one);
  one.l[0] = 1;
  BF(mul)(&t, a, &one);
  memcpy(b, t.l, FB);
}

#if defined(ORACLE_G2)
/* Fq2 = Fq[u] / (u^2 + 1): schoolbook / Karatsuba on the prime field above, under the names the rest of the file uses */
static ofq FQ_R1, FQ_ZERO; /* 1 and 0 of Fq2 */
static void fq_add(ofq* o, const ofq* a, const ofq* b) { fp_add(&o->c0, &a->c0, &b->c0); fp_add(&o->c1, &a->c1, &b->c1); }
static void fq_sub(ofq* o, const ofq* a, const ofq* b) { fp_sub(&o->c0, &a->c0, &b->c0); fp_sub(&o->c1, &a->c1, &b->c1); }
static void fq_neg(ofq* o, const ofq* a) { fp_neg(&o->c0, &a->c0); fp_neg(&o->c1, &a->c1); }
static void fq_dbl(ofq* o, const ofq* a) { fq_add(o, a, a); }
static int fq_is_zero(const ofq* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fq_eq(const ofq* a, const ofq* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fq_mul(ofq* o, const ofq* a, const ofq* b) {
  ofp v0, v1, s, t;
  fp_mul(&v0, &a->c0, &b->c0);
  fp_mul(&v1, &a->c1, &b->c1);
  fp_add(&s, &a->c0, &a->c1);
  fp_add(&t, &b->c0, &b->c1);
  fp_mul(&s, &s, &t);
  fp_sub(&s, &s, &v0);
  fp_sub(&o->c1, &s, &v1); /* a0 b1 + a1 b0 */
  fp_sub(&o->c0, &v0, &v1); /* a0 b0 - a1 b1 */
}
static void fq_sqr(ofq* o, const ofq* a) { fq_mul(o, a, a); }
static void fq_inv(ofq* o, const ofq* a) { /* (a0 - a1 u) / (a0^2 + a1^2) */
  ofp n, t;
  fp_sqr(&n, &a->c0);
  fp_sqr(&t, &a->c1);
  fp_add(&n, &n, &t);
  fp_inv(&n, &n);
  fp_mul(&o->c0, &a->c0, &n);
  fp_neg(&t, &a->c1);
  fp_mul(&o->c1, &t, &n);
}
static int fq_from_bytes(ofq* o, const uint8_t* b) {
  const int ok0 = fp_from_bytes(&o->c0, b), ok1 = fp_from_bytes(&o->c1, b + FB);
  return ok0 && ok1;
}
static void fq_to_bytes(uint8_t* b, const ofq* a) {
  fp_to_bytes(b, &a->c0);
  fp_to_bytes(b + FB, &a->c1);
}
#else
#define FQ_R1 BF_R1
#define FQ_R2 BF_R2
#define FQ_ZERO BF_ZERO
#endif
static ofq FQ_B3; /* the curve constant b (G2: the twist's), Montgomery form (the name is BN254 G1's: b = 3) */

static void oracle_init(void) {
  memset(&BF_ZERO, 0, sizeof BF_ZERO);
  /* R mod p by FBITS modular doublings of 1; R^2 by FBITS more */
  bfe one;
  memset(&one, 0, sizeof one);
  one.l[0] = 1;
  bfe t = one;
  for (int i = 0; i < FBITS; i++) BF(dbl)(&t, &t);
  BF_R1 = t;
  for (int i = 0; i < FBITS; i++) BF(dbl)(&t, &t);
  BF_R2 = t;
  bfe b = BF_ZERO; /* the G1 curve constant in Montgomery form (the name FQ_B3 is BN254's: b = 3) */
  for (int i = 0; i < CURVE_B_ABS; i++) BF(add)(&b, &b, &BF_R1);
  if (CURVE_B_IS_MINUS) BF(neg)(&b, &b);
#if defined(ORACLE_G2)
  memset(&FQ_ZERO, 0, sizeof FQ_ZERO);
  FQ_R1 = FQ_ZERO;
  FQ_R1.c0 = BF_R1;
  /* the twist's constant: b / xi with xi = 9 + u (BN254, a D-type twist: b' = 3 / (9 + u)), b * (1 + u) for BLS12-381 (M-type: 4 (1 + u)) */
  ofq bq = FQ_ZERO, xi = FQ_ZERO;
  bq.c0 = b;
#if defined(ORACLE_BLS12_381)
  xi.c0 = BF_R1;
  xi.c1 = BF_R1;
  fq_mul(&FQ_B3, &bq, &xi);
#else
  for (int i = 0; i < 9; i++) fp_add(&xi.c0, &xi.c0, &BF_R1);
  xi.c1 = BF_R1;
  fq_inv(&xi, &xi);
  fq_mul(&FQ_B3, &bq, &xi);
#endif
#else
  FQ_B3 = b;
#endif
}
static void ensure_init(void) { pthread_once(&g_once, oracle_init); }

/* ------------------------------------------------------------------------------------------------
 * G1 : y^2 = x^3 + 3, Jacobian coordinates.  Case split as src/cuzk/wgsl/curve/ec.template.wgsl:36-86
 * (add-2007-bl) and :10-34 (dbl-2009-l).
 * ---------------------------------------------------------------------------------------------- */
static void g1_identity(og1* o) {
  o->x = FQ_ZERO;
  o->y = FQ_R1;
  o->z = FQ_ZERO;
}
static int g1_is_identity(const og1* a) { return fq_is_zero(&a->z); }

static void g1_double(og1* o, const og1* p) {
  if (g1_is_identity(p)) {
    g1_identity(o);
    return;
  }
  ofq A, B, C, D, E, F, t, X3, Y3, Z3;
  fq_sqr(&A, &p->x);
  fq_sqr(&B, &p->y);
  fq_sqr(&C, &B);
  fq_add(&t, &p->x, &B);
  fq_sqr(&t, &t);
  fq_sub(&t, &t, &A);
  fq_sub(&t, &t, &C);
  fq_dbl(&D, &t);
  fq_dbl(&E, &A);
  fq_add(&E, &E, &A);
  fq_sqr(&F, &E);
  fq_dbl(&t, &D);
  fq_sub(&X3, &F, &t);
  fq_sub(&t, &D, &X3);
  fq_mul(&Y3, &E, &t);
  fq_dbl(&t, &C);
  fq_dbl(&t, &t);
  fq_dbl(&t, &t);
  fq_sub(&Y3, &Y3, &t);
  fq_mul(&Z3, &p->y, &p->z);
  fq_dbl(&Z3, &Z3);
  o->x = X3;
  o->y = Y3;
  o->z = Z3;
}

static void g1_add(og1* o, const og1* p, const og1* q) {
  if (g1_is_identity(p)) {
    *o = *q;
    return;
  }
  if (g1_is_identity(q)) {
    *o = *p;
    return;
  }
  ofq Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, r, V, t, X3, Y3, Z3;
  fq_sqr(&Z1Z1, &p->z);
  fq_sqr(&Z2Z2, &q->z);
  fq_mul(&U1, &p->x, &Z2Z2);
  fq_mul(&U2, &q->x, &Z1Z1);
  fq_mul(&S1, &p->y, &q->z);
  fq_mul(&S1, &S1, &Z2Z2);
  fq_mul(&S2, &q->y, &p->z);
  fq_mul(&S2, &S2, &Z1Z1);
  if (fq_eq(&U1, &U2)) {
    if (fq_eq(&S1, &S2)) {
      g1_double(o, p);
    } else {
      g1_identity(o);
    }
    return;
  }
  fq_sub(&H, &U2, &U1);
  fq_dbl(&I, &H);
  fq_sqr(&I, &I);
  fq_mul(&J, &H, &I);
  fq_sub(&r, &S2, &S1);
  fq_dbl(&r, &r);
  fq_mul(&V, &U1, &I);
  fq_sqr(&X3, &r);
  fq_sub(&X3, &X3, &J);
  fq_dbl(&t, &V);
  fq_sub(&X3, &X3, &t);
  fq_sub(&t, &V, &X3);
  fq_mul(&Y3, &r, &t);
  fq_mul(&t, &S1, &J);
  fq_dbl(&t, &t);
  fq_sub(&Y3, &Y3, &t);
  fq_add(&Z3, &p->z, &q->z);
  fq_sqr(&Z3, &Z3);
  fq_sub(&Z3, &Z3, &Z1Z1);
  fq_sub(&Z3, &Z3, &Z2Z2);
  fq_mul(&Z3, &Z3, &H);
  o->x = X3;
  o->y = Y3;
  o->z = Z3;
}

/* p (Jacobian) + (ax, ay) affine, optionally negated */
static void g1_add_affine(og1* o, const og1* p, const ofq* ax, const ofq* ay, int negate) {
  og1 q;
  q.x = *ax;
  if (negate)
    fq_neg(&q.y, ay);
  else
    q.y = *ay;
  q.z = FQ_R1;
  g1_add(o, p, &q);
}
static void g1_neg(og1* o, const og1* p) {
  o->x = p->x;
  fq_neg(&o->y, &p->y);
  o->z = p->z;
}
/* k as 32 LE bytes (any 256-bit integer) */
static void g1_mul_bytes(og1* o, const og1* p, const uint8_t k[32]) {
  og1 acc;
  g1_identity(&acc);
  for (int i = 255; i >= 0; i--) {
    g1_double(&acc, &acc);
    if ((k[i >> 3] >> (i & 7)) & 1) g1_add(&acc, &acc, p);
  }
  *o = acc;
}
static void g1_mul_u64(og1* o, const og1* p, uint64_t k) {
  uint8_t b[32] = {0};
  memcpy(b, &k, 8);
  g1_mul_bytes(o, p, b);
}
static void g1_from_bytes96(og1* o, const uint8_t* b) { /* (96: the record size of the 4-limb curves; JB in general) */
  fq_from_bytes(&o->x, b);
  fq_from_bytes(&o->y, b + CB);
  fq_from_bytes(&o->z, b + 2 * CB);
}
static void g1_to_bytes96(uint8_t* b, const og1* p) {
  fq_to_bytes(b, &p->x);
  fq_to_bytes(b + CB, &p->y);
  fq_to_bytes(b + 2 * CB, &p->z);
}

/* ------------------------------------------------------------------------------------------------ hooks */
void oracle_fq_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    ofq x, y, z;
    fq_from_bytes(&x, a + CB * i);
    if (b)
      fq_from_bytes(&y, b + CB * i);
    else
      y = FQ_ZERO;
    switch (op) {
      case 0: fq_add(&z, &x, &y); break;
      case 1: fq_sub(&z, &x, &y); break;
      case 2: fq_mul(&z, &x, &y); break;
      case 3: fq_sqr(&z, &x); break;
      case 4: fq_neg(&z, &x); break;
      default: fq_inv(&z, &x); break;
    }
    fq_to_bytes(out + CB * i, &z);
  }
}

void oracle_g1_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    og1 p, q, r;
    g1_from_bytes96(&p, a + JB * i);
    if (op == 0) {
      g1_from_bytes96(&q, b + JB * i);
      g1_add(&r, &p, &q);
    } else if (op == 1) {
      g1_double(&r, &p);
    } else {
      g1_neg(&r, &p);
    }
    g1_to_bytes96(out + JB * i, &r);
  }
}

void oracle_g1_scalar_mul(const uint8_t* p_xy, const uint8_t* k, uint8_t* out, size_t n) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    og1 p, r;
    fq_from_bytes(&p.x, p_xy + PB * i);
    fq_from_bytes(&p.y, p_xy + PB * i + CB);
    p.z = FQ_R1;
    g1_mul_bytes(&r, &p, k + 32 * i);
    g1_to_bytes96(out + JB * i, &r);
  }
}

int oracle_g1_to_affine64(const uint8_t* xyz, uint8_t* out) {
  ensure_init();
  og1 p;
  g1_from_bytes96(&p, xyz);
  if (g1_is_identity(&p)) {
    memset(out, 0, PB);
    return 1;
  }
  ofq zi, zi2, zi3, x, y;
  fq_inv(&zi, &p.z);
  fq_sqr(&zi2, &zi);
  fq_mul(&zi3, &zi2, &zi);
  fq_mul(&x, &p.x, &zi2);
  fq_mul(&y, &p.y, &zi3);
  fq_to_bytes(out, &x);
  fq_to_bytes(out + CB, &y);
  return 0;
}

int oracle_points_on_curve(const uint8_t* xy, size_t n) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    ofq x, y, l, r;
    if (!fq_from_bytes(&x, xy + PB * i)) return 0;
    if (!fq_from_bytes(&y, xy + PB * i + CB)) return 0;
    fq_sqr(&l, &y);
    fq_sqr(&r, &x);
    fq_mul(&r, &r, &x);
    fq_add(&r, &r, &FQ_B3);
    if (!fq_eq(&l, &r)) return 0;
  }
  return 1;
}

int oracle_coord_bytes(void) { return CB; }

void oracle_constants(uint8_t* p, uint8_t r[32], uint8_t* r2_mod_p, uint8_t* one_mont, uint64_t* n0inv64) {
  ensure_init();
  memcpy(p, FQ_P, FB); /* (prime-field constants in every build: FB bytes each) */
  memcpy(r, FR_R, 32);
  memcpy(r2_mod_p, BF_R2.l, FB);
  memcpy(one_mont, BF_R1.l, FB);
  *n0inv64 = FQ_N0;
}

/* ------------------------------------------------------------------------------------------------
 * MSM : restatement of halo2curves 0.9.0 `msm_serial` / `msm_best` (called at src/lib.rs:45-47).
 * Booth-recoded windows of c = ceil(ln n) bits (c = 1 for n < 4, 3 for n < 32), 2^(c-1) buckets per
 * window, summation by parts, windows combined from the top by c doublings.  (msm_best additionally
 * batches bucket additions in affine form for c >= 10; that changes the cost, not the group element.)
 * ---------------------------------------------------------------------------------------------- */
static unsigned scalar_bit(const uint8_t s[32], int bit) {
  if (bit < 0 || bit >= 256) return 0;
  return (s[bit >> 3] >> (bit & 7)) & 1u;
}
/* signed digit of window `win`: slice of c+1 bits whose lowest bit is bit (win*c - 1), bit -1 := 0 */
static int32_t booth_digit(int win, int c, const uint8_t s[32]) {
  uint32_t u = 0;
  int lo = win * c - 1;
  for (int k = c; k >= 0; k--) u = (u << 1) | scalar_bit(s, lo + k);
  int32_t t = (int32_t)((u + 1) >> 1);
  if (u >> c) t -= (int32_t)1 << c;
  return t;
}
static int msm_window_bits(size_t n) {
  if (n < 4) return 1;
  if (n < 32) return 3;
  return (int)ceil(log((double)n));
}

static void msm_serial(const ofq* bx, const ofq* by, const uint8_t* scalars, size_t n, og1* acc) {
  int c = msm_window_bits(n);
  int num_bits = 256; /* Fr::NUM_BITS: the bit length of the scalar modulus (254 for BN254 / Grumpkin, 255 for Pallas / Vesta) */
  while (!((FR_R[(num_bits - 1) / 64] >> ((num_bits - 1) % 64)) & 1)) num_bits--;
  int nwin = num_bits / c + 1;
  size_t nb = (size_t)1 << (c - 1);
  og1* buckets = (og1*)malloc(nb * sizeof(og1));
  g1_identity(acc);
  for (int w = nwin - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) g1_double(acc, acc);
    for (size_t b = 0; b < nb; b++) g1_identity(&buckets[b]);
    for (size_t i = 0; i < n; i++) {
      int32_t d = booth_digit(w, c, scalars + 32 * i);
      if (d > 0)
        g1_add_affine(&buckets[d - 1], &buckets[d - 1], &bx[i], &by[i], 0);
      else if (d < 0)
        g1_add_affine(&buckets[-d - 1], &buckets[-d - 1], &bx[i], &by[i], 1);
    }
    og1 running;
    g1_identity(&running);
    for (size_t b = nb; b-- > 0;) {
      g1_add(&running, &running, &buckets[b]);
      g1_add(acc, acc, &running);
    }
  }
  free(buckets);
}

typedef struct {
  const ofq *bx, *by;
  const uint8_t* scalars;
  size_t n;
  og1 out;
} msm_job;
static void* msm_job_run(void* arg) {
  msm_job* j = (msm_job*)arg;
  msm_serial(j->bx, j->by, j->scalars, j->n, &j->out);
  return NULL;
}

int oracle_msm_bn254_g1_mt(const uint8_t* xy, const uint8_t* scalars, size_t n, int n_threads, uint8_t* out_xyz) {
  ensure_init();
  og1 acc;
  g1_identity(&acc);
  if (n == 0) {
    g1_to_bytes96(out_xyz, &acc);
    return 0;
  }
  ofq* bx = (ofq*)malloc(n * sizeof(ofq));
  ofq* by = (ofq*)malloc(n * sizeof(ofq));
  for (size_t i = 0; i < n; i++) {
    fq_from_bytes(&bx[i], xy + PB * i);
    fq_from_bytes(&by[i], xy + PB * i + CB);
  }
  if (n_threads < 1) n_threads = 1;
  if ((size_t)n_threads > n) n_threads = (int)n;
  if (n_threads == 1) {
    msm_serial(bx, by, scalars, n, &acc);
  } else {
    msm_job* jobs = (msm_job*)calloc((size_t)n_threads, sizeof(msm_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    size_t per = (n + (size_t)n_threads - 1) / (size_t)n_threads;
    int used = 0;
    for (int t = 0; t < n_threads; t++) {
      size_t lo = (size_t)t * per;
      if (lo >= n) break;
      size_t hi = lo + per < n ? lo + per : n;
      jobs[t].bx = bx + lo;
      jobs[t].by = by + lo;
      jobs[t].scalars = scalars + 32 * lo;
      jobs[t].n = hi - lo;
      pthread_create(&th[t], NULL, msm_job_run, &jobs[t]);
      used++;
    }
    for (int t = 0; t < used; t++) {
      pthread_join(th[t], NULL);
      g1_add(&acc, &acc, &jobs[t].out);
    }
    free(jobs);
    free(th);
  }
  free(bx);
  free(by);
  g1_to_bytes96(out_xyz, &acc);
  return 0;
}

int oracle_msm_bn254_g1(const uint8_t* xy, const uint8_t* scalars, size_t n, uint8_t* out_xyz) {
  return oracle_msm_bn254_g1_mt(xy, scalars, n, 1, out_xyz);
}

/* ------------------------------------------------------------------------------------------------
 * cuZK stage models
 * ---------------------------------------------------------------------------------------------- */
int oracle_decompose_scalars_signed(const uint8_t* scalars, size_t n, int num_words, int word_size, int32_t* digits) {
  const int32_t l = (int32_t)1 << word_size, h = l >> 1;
  for (size_t i = 0; i < n; i++) {
    const uint8_t* s = scalars + 32 * i;
    int32_t carry = 0;
    for (int w = 0; w < num_words; w++) {
      int32_t raw = 0;
      for (int k = word_size - 1; k >= 0; k--) raw = (raw << 1) | (int32_t)scalar_bit(s, w * word_size + k);
      int32_t d = raw + carry;
      if (d >= h) {
        d -= l;
        carry = 1;
      } else {
        carry = 0;
      }
      digits[(size_t)w * n + i] = d + h;
    }
    if (carry) return -1; /* src/cuzk/test/utils.rs:150-152 panics */
  }
  return 0;
}

void oracle_transpose(const int32_t* digits_w, size_t n, int num_columns, int32_t* col_ptr, int32_t* val_idxs) {
  memset(col_ptr, 0, sizeof(int32_t) * ((size_t)num_columns + 1));
  for (size_t i = 0; i < n; i++) col_ptr[digits_w[i] + 1]++;
  for (int b = 0; b < num_columns; b++) col_ptr[b + 1] += col_ptr[b];
  int32_t* cur = (int32_t*)calloc((size_t)num_columns, sizeof(int32_t));
  for (size_t i = 0; i < n; i++) {
    int32_t d = digits_w[i];
    val_idxs[col_ptr[d] + cur[d]] = (int32_t)i;
    cur[d]++;
  }
  free(cur);
}

void oracle_smvp_signed(const int32_t* col_ptr, const int32_t* val_idxs, const uint8_t* xy, size_t n, int num_columns,
                        uint8_t* buckets_xyz) {
  ensure_init();
  (void)n;
  const int h = num_columns / 2;
  for (int k = 0; k < h; k++) {
    og1 bucket;
    g1_identity(&bucket);
    for (int j = 0; j < 2; j++) {
      int row = (j == 0) ? k + h : h - k;
      if (k == 0 && j == 0) row = 0;
      og1 sum;
      g1_identity(&sum);
      for (int32_t t = col_ptr[row]; t < col_ptr[row + 1]; t++) {
        ofq x, y;
        const uint8_t* pt = xy + PB * (size_t)val_idxs[t];
        fq_from_bytes(&x, pt);
        fq_from_bytes(&y, pt + CB);
        g1_add_affine(&sum, &sum, &x, &y, 0);
      }
      int bi;
      if (h > row) {
        bi = h - row;
        g1_neg(&sum, &sum);
      } else {
        bi = row - h;
      }
      if (bi > 0) g1_add(&bucket, &bucket, &sum);
    }
    g1_to_bytes96(buckets_xyz + JB * (size_t)k, &bucket);
  }
}

static og1* load_points96(const uint8_t* b, int n) {
  og1* p = (og1*)malloc(sizeof(og1) * (size_t)n);
  for (int i = 0; i < n; i++) g1_from_bytes96(&p[i], b + JB * (size_t)i);
  return p;
}

static void par_reduce_1(const og1* buckets, int nb, int nt, og1* g_out, og1* m_out) {
  int per = nb / nt;
  for (int t = 0; t < nt; t++) {
    int idx = (t == 0) ? 0 : (nt - t) * per;
    og1 m = buckets[idx], g = m;
    for (int i = 0; i < per - 1; i++) {
      g1_add(&m, &m, &buckets[(nt - t) * per - 1 - i]);
      g1_add(&g, &g, &m);
    }
    g_out[t] = g;
    m_out[t] = m;
  }
}
static void par_reduce_2(const og1* g_in, const og1* m_in, int nb, int nt, og1* out) {
  int per = nb / nt;
  for (int t = 0; t < nt; t++) {
    og1 g = g_in[t];
    uint64_t s = (uint64_t)per * (uint64_t)(nt - t - 1);
    if (s > 0) {
      og1 ms;
      g1_mul_u64(&ms, &m_in[t], s);
      g1_add(&g, &g, &ms);
    }
    out[t] = g;
  }
}

void oracle_bucket_reduction(int kind, const uint8_t* buckets_xyz, int num_buckets, int num_threads, uint8_t* out_xyz) {
  ensure_init();
  og1* b = load_points96(buckets_xyz, num_buckets);
  og1 acc;
  g1_identity(&acc);
  if (kind == 0) { /* serial_bucket_reduction: indices 1..h-1 then 0 with weights 1..h */
    for (int i = 1; i <= num_buckets; i++) {
      int idx = (i < num_buckets) ? i : 0;
      og1 t;
      g1_mul_u64(&t, &b[idx], (uint64_t)i);
      g1_add(&acc, &acc, &t);
    }
  } else if (kind == 1) { /* running_sum_bucket_reduction */
    og1 m = b[0], g = m;
    for (int i = 0; i < num_buckets - 1; i++) {
      g1_add(&m, &m, &b[num_buckets - 1 - i]);
      g1_add(&g, &g, &m);
    }
    acc = g;
  } else {
    og1* g = (og1*)malloc(sizeof(og1) * (size_t)num_threads);
    og1* m = (og1*)malloc(sizeof(og1) * (size_t)num_threads);
    og1* r = (og1*)malloc(sizeof(og1) * (size_t)num_threads);
    par_reduce_1(b, num_buckets, num_threads, g, m);
    par_reduce_2(g, m, num_buckets, num_threads, r);
    for (int t = 0; t < num_threads; t++) g1_add(&acc, &acc, &r[t]);
    free(g);
    free(m);
    free(r);
  }
  g1_to_bytes96(out_xyz, &acc);
  free(b);
}

void oracle_parallel_bucket_reduction_1(const uint8_t* buckets_xyz, int num_buckets, int num_threads, uint8_t* g_out,
                                        uint8_t* m_out) {
  ensure_init();
  og1* b = load_points96(buckets_xyz, num_buckets);
  og1* g = (og1*)malloc(sizeof(og1) * (size_t)num_threads);
  og1* m = (og1*)malloc(sizeof(og1) * (size_t)num_threads);
  par_reduce_1(b, num_buckets, num_threads, g, m);
  for (int t = 0; t < num_threads; t++) {
    g1_to_bytes96(g_out + JB * (size_t)t, &g[t]);
    g1_to_bytes96(m_out + JB * (size_t)t, &m[t]);
  }
  free(b);
  free(g);
  free(m);
}

void oracle_parallel_bucket_reduction_2(const uint8_t* g_in, const uint8_t* m_in, int num_buckets, int num_threads,
                                        uint8_t* out) {
  ensure_init();
  og1* g = load_points96(g_in, num_threads);
  og1* m = load_points96(m_in, num_threads);
  og1* r = (og1*)malloc(sizeof(og1) * (size_t)num_threads);
  par_reduce_2(g, m, num_buckets, num_threads, r);
  for (int t = 0; t < num_threads; t++) g1_to_bytes96(out + JB * (size_t)t, &r[t]);
  free(g);
  free(m);
  free(r);
}

void oracle_horner(const uint8_t* window_sums_xyz, int num_words, int word_size, uint8_t* out_xyz) {
  ensure_init();
  og1* s = load_points96(window_sums_xyz, num_words);
  og1 acc = s[num_words - 1];
  for (int w = num_words - 2; w >= 0; w--) {
    for (int k = 0; k < word_size; k++) g1_double(&acc, &acc);
    g1_add(&acc, &acc, &s[w]);
  }
  g1_to_bytes96(out_xyz, &acc);
  free(s);
}

int oracle_msm_cuzk_model(const uint8_t* xy, const uint8_t* scalars, size_t n, int word_size, uint8_t* out_xyz) {
  ensure_init();
  int num_words = (256 + word_size - 1) / word_size;
  int num_columns = 1 << word_size;
  int h = num_columns / 2;
  int32_t* digits = (int32_t*)malloc(sizeof(int32_t) * (size_t)num_words * (n ? n : 1));
  if (oracle_decompose_scalars_signed(scalars, n, num_words, word_size, digits) != 0) {
    free(digits);
    return -1;
  }
  int32_t* col_ptr = (int32_t*)malloc(sizeof(int32_t) * ((size_t)num_columns + 1));
  int32_t* val = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  uint8_t* buckets = (uint8_t*)malloc(JB * (size_t)h);
  uint8_t* sums = (uint8_t*)malloc(JB * (size_t)num_words);
  for (int w = 0; w < num_words; w++) {
    oracle_transpose(digits + (size_t)w * n, n, num_columns, col_ptr, val);
    oracle_smvp_signed(col_ptr, val, xy, n, num_columns, buckets);
    oracle_bucket_reduction(1, buckets, h, 1, sums + JB * (size_t)w);
  }
  oracle_horner(sums, num_words, word_size, out_xyz);
  free(digits);
  free(col_ptr);
  free(val);
  free(buckets);
  free(sums);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * deterministic synthetic inputs -- identical definition in oracle/bn254_ref.py
 * ---------------------------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* nl 64-bit words: 4 = a scalar or a coordinate of the 4-limb fields, masked to 254 bits; 6 = a BLS12-381 coordinate, masked to p's 381 bits */
static void draw_words(uint64_t seed, uint64_t index, uint64_t attempt, uint64_t domain, uint64_t* out, int nl) {
  uint64_t base = splitmix64(seed ^ ((domain & 0xFF) << 56)) ^ (index * 0xD1342543DE82EF95ull);
  base = splitmix64(base ^ (attempt * 0xA0761D6478BD642Full));
  uint64_t s = base;
  for (int i = 0; i < nl; i++) {
    s = splitmix64(s);
    out[i] = s;
  }
  if (nl == 4) out[3] &= 0x3FFFFFFFFFFFFFFFull; /* 254 bits */
  else out[nl - 1] &= (1ull << (381 - 64 * (nl - 1))) - 1; /* 381 bits */
}
static void draw256(uint64_t seed, uint64_t index, uint64_t attempt, uint64_t domain, uint64_t out[4]) { draw_words(seed, index, attempt, domain, out, 4); }
static __attribute__((unused)) int ltn(const uint64_t* a, const uint64_t* m, int nl) {
  for (int i = nl - 1; i >= 0; i--) {
    if (a[i] < m[i]) return 1;
    if (a[i] > m[i]) return 0;
  }
  return 0;
}
static int lt4(const uint64_t a[4], const uint64_t m[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] < m[i]) return 1;
    if (a[i] > m[i]) return 0;
  }
  return 0;
}

void oracle_sample_scalars(uint64_t seed, size_t first, size_t n, uint8_t* out32) {
  for (size_t i = 0; i < n; i++) {
    uint64_t v[4];
    for (uint64_t attempt = 0;; attempt++) {
      draw256(seed, first + i, attempt, 1, v);
      if (lt4(v, FR_R)) break;
    }
    memcpy(out32 + 32 * i, v, 32);
  }
}

#if !defined(ORACLE_G2)
/* square root in Fq for the samplers: y with y^2 = a, or 0 (return value) if a is not a square.
 * p = 3 mod 4 (BN254 Fq): a^((p+1)/4).  Otherwise (Grumpkin's base field, p - 1 = 2^28 t): Tonelli-Shanks. */
static int fq_sqrt(ofq* y, const ofq* a, const uint64_t* e_p3) {
  ofq y2;
  if ((FQ_P[0] & 3) == 3) {
    fq_pow(y, a, e_p3);
  } else {
    /* p - 1 = 2^s t */
    uint64_t t[NL];
    memcpy(t, FQ_P, CB);
    t[0] -= 1;
    int s = 0;
    while (!(t[0] & 1)) {
      for (int i = 0; i < NL; i++) t[i] = (t[i] >> 1) | (i < NL - 1 ? t[i + 1] << 63 : 0);
      s++;
    }
    uint64_t half[NL]; /* (p - 1) / 2 */
    {
      uint64_t pm1[NL];
      memcpy(pm1, FQ_P, CB);
      pm1[0] -= 1;
      for (int i = 0; i < NL; i++) half[i] = (pm1[i] >> 1) | (i < NL - 1 ? pm1[i + 1] << 63 : 0);
    }
    static ofq c0; /* z^t for the smallest non-residue z: a generator of the 2-Sylow subgroup */
    static int have_c0 = 0;
    if (!have_c0) {
      ofq z = FQ_R1, e;
      for (;;) {
        fq_add(&z, &z, &FQ_R1); /* 2, 3, ... */
        fq_pow(&e, &z, half);
        if (!fq_eq(&e, &FQ_R1)) break;
      }
      fq_pow(&c0, &z, t);
      have_c0 = 1;
    }
    ofq chk;
    fq_pow(&chk, a, half);
    if (!fq_eq(&chk, &FQ_R1) && !fq_is_zero(a)) return 0;
    uint64_t tp1h[NL]; /* (t + 1) / 2 */
    {
      u128 c = (u128)t[0] + 1;
      uint64_t u[NL];
      u[0] = (uint64_t)c;
      c >>= 64;
      for (int i = 1; i < NL; i++) {
        c += t[i];
        u[i] = (uint64_t)c;
        c >>= 64;
      }
      for (int i = 0; i < NL; i++) tp1h[i] = (u[i] >> 1) | (i < NL - 1 ? u[i + 1] << 63 : 0);
    }
    ofq x, b, c = c0;
    fq_pow(&x, a, tp1h);
    fq_pow(&b, a, t);
    int m = s;
    while (!fq_eq(&b, &FQ_R1) && !fq_is_zero(&b)) {
      int i = 0;
      ofq q = b;
      while (!fq_eq(&q, &FQ_R1)) {
        fq_sqr(&q, &q);
        i++;
      }
      ofq bb = c;
      for (int k = 0; k < m - i - 1; k++) fq_sqr(&bb, &bb);
      fq_mul(&x, &x, &bb);
      fq_sqr(&c, &bb);
      fq_mul(&b, &b, &c);
      m = i;
    }
    *y = x;
  }
  fq_sqr(&y2, y);
  return fq_eq(&y2, a);
}

void oracle_sample_points(uint64_t seed, size_t first, size_t n, uint8_t* out64) {
  ensure_init();
  /* (p + 1) / 4 */
  uint64_t e[NL];
  {
    u128 c = (u128)FQ_P[0] + 1;
    uint64_t t[NL];
    t[0] = (uint64_t)c;
    c >>= 64;
    for (int i = 1; i < NL; i++) {
      c += FQ_P[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
    for (int i = 0; i < NL; i++) e[i] = (t[i] >> 2) | (i < NL - 1 ? t[i + 1] << 62 : 0);
  }
  for (size_t i = 0; i < n; i++) {
    for (uint64_t attempt = 0;; attempt++) {
      uint64_t v[NL];
      draw_words(seed, first + i, attempt, 2, v, NL);
      if (!ltn(v, FQ_P, NL)) continue;
      ofq x, rhs, y, y2;
      uint8_t xb[CB];
      memcpy(xb, v, CB);
      fq_from_bytes(&x, xb);
      fq_sqr(&rhs, &x);
      fq_mul(&rhs, &rhs, &x);
      fq_add(&rhs, &rhs, &FQ_B3);
      (void)y2;
      if (!fq_sqrt(&y, &rhs, e)) continue;
      uint8_t yb[CB];
      fq_to_bytes(yb, &y);
      if ((unsigned)(yb[0] & 1) != (unsigned)((xb[0] >> 1) & 1)) {
        fq_neg(&y, &y);
        fq_to_bytes(yb, &y);
      }
      memcpy(out64 + PB * i, xb, CB);
      memcpy(out64 + PB * i + CB, yb, CB);
      break;
    }
  }
}
#else /* ORACLE_G2 */
/* G2 has no try-and-increment sampler here (a square root in Fq2): the synthetic points are KNOWN multiples of the generator,
 * P_i = (a + i b) G with a = sample_scalar(seed ^ 0x6732, 0) | 1, b = sample_scalar(seed ^ 0x6732, 1) | 1 -- the definition of
 * oracle/bn254_g2_ref.py: sample_points, which this must reproduce byte for byte -- by repeated addition and one batched inversion. */
static const uint64_t G2_GEN[4][NL] = {
#if defined(ORACLE_BLS12_381) /* the standard generator of BLS12-381's G2 (x.c0, x.c1, y.c0, y.c1) */
  {0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull},
  {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull},
  {0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull},
  {0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull},
#else /* BN254: the generator of EIP-197 */
  {0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull},
  {0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull},
  {0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull},
  {0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull},
#endif
};
void oracle_sample_points(uint64_t seed, size_t first, size_t n, uint8_t* out64) {
  ensure_init();
  if (n == 0) return;
  uint8_t ab[64], gen[PB];
  oracle_sample_scalars(seed ^ 0x6732ull, 0, 2, ab);
  ab[0] |= 1;
  ab[32] |= 1;
  memcpy(gen, G2_GEN, PB);
  og1 g, cur, step, t;
  fq_from_bytes(&g.x, gen);
  fq_from_bytes(&g.y, gen + CB);
  g.z = FQ_R1;
  g1_mul_bytes(&cur, &g, ab);        /* a G */
  g1_mul_bytes(&step, &g, ab + 32);  /* b G */
  g1_mul_u64(&t, &step, (uint64_t)first);
  g1_add(&cur, &cur, &t);
  og1* jac = (og1*)malloc(n * sizeof(og1));
  ofq* pref = (ofq*)malloc(n * sizeof(ofq));
  ofq run = FQ_R1;
  for (size_t i = 0; i < n; i++) {  /* (no point of the sequence is the identity: a + i b = 0 mod r has probability ~2^-234) */
    jac[i] = cur;
    pref[i] = run;
    fq_mul(&run, &run, &cur.z);
    g1_add(&cur, &cur, &step);
  }
  ofq inv;
  fq_inv(&inv, &run);
  for (size_t i = n; i-- > 0;) {
    ofq zi, zi2, zi3, x, y;
    fq_mul(&zi, &inv, &pref[i]);
    fq_mul(&inv, &inv, &jac[i].z);
    fq_sqr(&zi2, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(&x, &jac[i].x, &zi2);
    fq_mul(&y, &jac[i].y, &zi3);
    fq_to_bytes(out64 + PB * i, &x);
    fq_to_bytes(out64 + PB * i + CB, &y);
  }
  free(jac);
  free(pref);
}
#endif
