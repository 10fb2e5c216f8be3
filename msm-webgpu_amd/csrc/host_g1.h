// Host finalisation arithmetic: the unit's base field as HL x 64-bit Montgomery limbs (4, R = 2^256, for BN254 and the other 254 / 255-bit
// fields; 6 for BLS12-381) and Jacobian G1.
//
// The only host-side group arithmetic in the product: the window combine result = sum_w 2^(16w) S_w
// (≙ src/cuzk/msm.rs:391-416, which the reference also runs on the host).  It is 240 dependent doublings --
// a serial chain that one CPU core finishes in ~60 us while a single GPU lane would need milliseconds --
// so it stays on the host by design (DESIGN.md "Window combine").  Not a fallback for any device stage.
#ifndef MSM_CURVE_UNIT
#pragma once
#include "curve_select.h"
#include MSM_CURVE_CONSTANTS
#endif
#include <cstdint>
#include <cstring>

namespace MSM_FIELD_NS {
namespace host {

typedef unsigned __int128 u128;

// HL 64-bit limbs per prime-field element (4 for the 254 / 255-bit fields, 6 for BLS12-381), CB bytes per coordinate on the wire (32 / 48;
// 64 in the G2 unit, whose coordinates are Fq2 elements c0 || c1); a Jacobian record is 3 CB bytes, an affine one 2 CB
constexpr int HL = FQ_WORDS / 2 / FQ_EXT;
constexpr int CB = 4 * FQ_WORDS;
constexpr int JB = 3 * CB;
constexpr int PCB = CB / FQ_EXT;  // bytes of a prime-field element

// the prime field (in a G2 unit: `hfp`, with the coordinate field `hfq` = Fq2 defined on it below)
#if defined(MSM_FQ2)
#define HFQ_BASE(name) hfp##name
#else
#define HFQ_BASE(name) hfq##name
#endif
#define hfq_t HFQ_BASE()

struct hfq_t {
  uint64_t l[HL];
};

inline bool HFQ_BASE(_geq_p)(const hfq_t& a) {
  for (int i = HL - 1; i >= 0; i--) {
    if (a.l[i] != FQ_P64[i]) return a.l[i] > FQ_P64[i];
  }
  return true;
}
inline void HFQ_BASE(_sub_p)(hfq_t& a) {
  uint64_t borrow = 0;
  for (int i = 0; i < HL; i++) {
    u128 d = (u128)a.l[i] - FQ_P64[i] - borrow;
    a.l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1u;
  }
}
inline hfq_t HFQ_BASE(_add)(const hfq_t& a, const hfq_t& b) {
  hfq_t r;
  u128 c = 0;
  for (int i = 0; i < HL; i++) {
    c += (u128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  if (HFQ_BASE(_geq_p)(r)) HFQ_BASE(_sub_p)(r);
  return r;
}
inline hfq_t HFQ_BASE(_sub)(const hfq_t& a, const hfq_t& b) {
  hfq_t r;
  uint64_t borrow = 0;
  for (int i = 0; i < HL; i++) {
    u128 d = (u128)a.l[i] - b.l[i] - borrow;
    r.l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1u;
  }
  if (borrow) {
    u128 c = 0;
    for (int i = 0; i < HL; i++) {
      c += (u128)r.l[i] + FQ_P64[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
// Montgomery product (R = 2^(64 HL)), separated operand scanning: the full double-width product first, then HL reduction rounds
inline hfq_t HFQ_BASE(_mul)(const hfq_t& a, const hfq_t& b) {
  uint64_t t[2 * HL + 1];
  for (int i = 0; i < 2 * HL + 1; i++) t[i] = 0;
  for (int i = 0; i < HL; i++) {
    u128 carry = 0;
    for (int j = 0; j < HL; j++) {
      carry += (u128)a.l[i] * b.l[j] + t[i + j];
      t[i + j] = (uint64_t)carry;
      carry >>= 64;
    }
    t[i + HL] = (uint64_t)carry;
  }
  for (int i = 0; i < HL; i++) {
    const uint64_t m = t[i] * FQ_N0_64;
    u128 carry = 0;
    for (int j = 0; j < HL; j++) {
      carry += (u128)m * FQ_P64[j] + t[i + j];
      t[i + j] = (uint64_t)carry;
      carry >>= 64;
    }
    for (int k = i + HL; carry && k < 2 * HL + 1; k++) {
      carry += t[k];
      t[k] = (uint64_t)carry;
      carry >>= 64;
    }
  }
  hfq_t r;
  for (int i = 0; i < HL; i++) r.l[i] = t[HL + i];
  if (t[2 * HL] || HFQ_BASE(_geq_p)(r)) HFQ_BASE(_sub_p)(r);
  return r;
}
inline hfq_t HFQ_BASE(_sqr)(const hfq_t& a) { return HFQ_BASE(_mul)(a, a); }
inline bool HFQ_BASE(_is_zero)(const hfq_t& a) {
  uint64_t z = 0;
  for (int i = 0; i < HL; i++) z |= a.l[i];
  return z == 0;
}
inline bool HFQ_BASE(_eq)(const hfq_t& a, const hfq_t& b) { return memcmp(a.l, b.l, PCB) == 0; }
inline hfq_t HFQ_BASE(_const)(const uint64_t* c) {
  hfq_t r;
  for (int i = 0; i < HL; i++) r.l[i] = c[i];
  return r;
}

// canonical little-endian bytes <-> Montgomery; returns false when the encoding is >= p
inline bool HFQ_BASE(_from_bytes)(hfq_t& r, const uint8_t* b) {
  hfq_t t;
  memcpy(t.l, b, PCB);
  const bool ok = !HFQ_BASE(_geq_p)(t);
  r = HFQ_BASE(_mul)(t, HFQ_BASE(_const)(FQ_R2_64));
  return ok;
}
inline void HFQ_BASE(_to_bytes)(uint8_t* b, const hfq_t& a) {
  hfq_t one;
  for (int i = 0; i < HL; i++) one.l[i] = i == 0 ? 1 : 0;
  hfq_t t = HFQ_BASE(_mul)(a, one);
  memcpy(b, t.l, PCB);
}

inline hfq_t HFQ_BASE(_inv)(const hfq_t& a) {  // a^(p-2), square-and-multiply from the top bit
  uint64_t e[HL];
  for (int i = 0; i < HL; i++) e[i] = FQ_P64[i];
  e[0] -= 2;  // (p is odd and > 2: no borrow)
  hfq_t acc = HFQ_BASE(_const)(FQ_ONE64);
  for (int i = 64 * HL - 1; i >= 0; i--) {
    acc = HFQ_BASE(_sqr)(acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = HFQ_BASE(_mul)(acc, a);
  }
  return acc;
}

#undef hfq_t
#if defined(MSM_FQ2)
// The coordinate field of a G2 unit: Fq2 = Fq[u] / (u^2 + 1) on the prime field above, under the names the group code below uses.
// Wire form of an element: c0 || c1, each PCB bytes canonical little-endian.
struct hfq {
  hfp c0, c1;
};
inline hfq hfq_add(const hfq& a, const hfq& b) { return {hfp_add(a.c0, b.c0), hfp_add(a.c1, b.c1)}; }
inline hfq hfq_sub(const hfq& a, const hfq& b) { return {hfp_sub(a.c0, b.c0), hfp_sub(a.c1, b.c1)}; }
inline hfq hfq_mul(const hfq& a, const hfq& b) {  // Karatsuba: 3 prime-field products
  const hfp v0 = hfp_mul(a.c0, b.c0), v1 = hfp_mul(a.c1, b.c1);
  const hfp s = hfp_mul(hfp_add(a.c0, a.c1), hfp_add(b.c0, b.c1));
  return {hfp_sub(v0, v1), hfp_sub(hfp_sub(s, v0), v1)};
}
inline hfq hfq_sqr(const hfq& a) {
  const hfp t = hfp_mul(a.c0, a.c1);
  return {hfp_mul(hfp_add(a.c0, a.c1), hfp_sub(a.c0, a.c1)), hfp_add(t, t)};
}
inline bool hfq_is_zero(const hfq& a) { return hfp_is_zero(a.c0) && hfp_is_zero(a.c1); }
inline bool hfq_eq(const hfq& a, const hfq& b) { return hfp_eq(a.c0, b.c0) && hfp_eq(a.c1, b.c1); }
inline bool hfq_from_bytes(hfq& r, const uint8_t* b) {
  const bool ok0 = hfp_from_bytes(r.c0, b), ok1 = hfp_from_bytes(r.c1, b + PCB);
  return ok0 && ok1;
}
inline void hfq_to_bytes(uint8_t* b, const hfq& a) {
  hfp_to_bytes(b, a.c0);
  hfp_to_bytes(b + PCB, a.c1);
}
inline hfq hfq_inv(const hfq& a) {  // 1 / (c0 + c1 u) = (c0 - c1 u) / (c0^2 + c1^2)
  const hfp n = hfp_inv(hfp_add(hfp_sqr(a.c0), hfp_sqr(a.c1)));
  hfp zero;
  for (int i = 0; i < HL; i++) zero.l[i] = 0;
  return {hfp_mul(a.c0, n), hfp_mul(hfp_sub(zero, a.c1), n)};
}
#endif
#undef HFQ_BASE

struct hg1 {  // Jacobian, z == 0 <=> identity
  hfq x, y, z;
};
inline hg1 hg1_identity() {
  hg1 r;
  memset(&r, 0, sizeof r);
  return r;
}
inline bool hg1_is_identity(const hg1& p) { return hfq_is_zero(p.z); }
inline hg1 hg1_neg(const hg1& p) {  // (x, -y, z): 0 - y in the coordinate field (prime field or Fq2)
  hg1 zero = hg1_identity();
  return {p.x, hfq_sub(zero.y, p.y), p.z};
}

inline hg1 hg1_double(const hg1& p) {  // dbl-2009-l, a = 0
  if (hg1_is_identity(p)) return p;
  hfq A = hfq_sqr(p.x), B = hfq_sqr(p.y), C = hfq_sqr(B);
  hfq t = hfq_add(p.x, B);
  t = hfq_sub(hfq_sub(hfq_sqr(t), A), C);
  hfq D = hfq_add(t, t);
  hfq E = hfq_add(hfq_add(A, A), A);
  hfq F = hfq_sqr(E);
  hg1 r;
  r.x = hfq_sub(F, hfq_add(D, D));
  hfq c8 = hfq_add(C, C);
  c8 = hfq_add(c8, c8);
  c8 = hfq_add(c8, c8);
  r.y = hfq_sub(hfq_mul(E, hfq_sub(D, r.x)), c8);
  hfq yz = hfq_mul(p.y, p.z);
  r.z = hfq_add(yz, yz);
  return r;
}
inline hg1 hg1_add(const hg1& p, const hg1& q) {  // add-2007-bl with the usual case split
  if (hg1_is_identity(p)) return q;
  if (hg1_is_identity(q)) return p;
  hfq z1z1 = hfq_sqr(p.z), z2z2 = hfq_sqr(q.z);
  hfq u1 = hfq_mul(p.x, z2z2), u2 = hfq_mul(q.x, z1z1);
  hfq s1 = hfq_mul(hfq_mul(p.y, q.z), z2z2), s2 = hfq_mul(hfq_mul(q.y, p.z), z1z1);
  if (hfq_eq(u1, u2)) return hfq_eq(s1, s2) ? hg1_double(p) : hg1_identity();
  hfq h = hfq_sub(u2, u1);
  hfq i = hfq_add(h, h);
  i = hfq_sqr(i);
  hfq j = hfq_mul(h, i);
  hfq rr = hfq_sub(s2, s1);
  rr = hfq_add(rr, rr);
  hfq v = hfq_mul(u1, i);
  hg1 r;
  r.x = hfq_sub(hfq_sub(hfq_sqr(rr), j), hfq_add(v, v));
  hfq s1j = hfq_mul(s1, j);
  r.y = hfq_sub(hfq_mul(rr, hfq_sub(v, r.x)), hfq_add(s1j, s1j));
  hfq zz = hfq_add(p.z, q.z);
  r.z = hfq_mul(hfq_sub(hfq_sub(hfq_sqr(zz), z1z1), z2z2), h);
  return r;
}
// (the "96" of these names is the record size of the 254 / 255-bit curves; a record is JB = 3 CB bytes)
inline bool hg1_from_bytes96(hg1& r, const uint8_t* b) {
  bool ok = hfq_from_bytes(r.x, b);
  ok &= hfq_from_bytes(r.y, b + CB);
  ok &= hfq_from_bytes(r.z, b + 2 * CB);
  return ok;
}
inline void hg1_to_bytes96(uint8_t* b, const hg1& p) {
  if (hg1_is_identity(p)) {
    memset(b, 0, JB);
    return;
  }
  hfq_to_bytes(b, p.x);
  hfq_to_bytes(b + CB, p.y);
  hfq_to_bytes(b + 2 * CB, p.z);
}

// Jacobian bytes -> canonical affine x || y (≙ Curve::to_affine); returns 1 for the identity (out zeroed), -1 on a
// non-canonical coordinate, 0 otherwise
inline int to_affine64(const uint8_t* xyz, uint8_t* out) {
  hg1 p;
  if (!hg1_from_bytes96(p, xyz)) return -1;
  if (hg1_is_identity(p)) {
    memset(out, 0, 2 * CB);
    return 1;
  }
  const hfq zi = hfq_inv(p.z), zi2 = hfq_sqr(zi);
  hfq_to_bytes(out, hfq_mul(p.x, zi2));
  hfq_to_bytes(out + CB, hfq_mul(p.y, hfq_mul(zi2, zi)));
  return 0;
}

// result = sum_w 2^(window_bits * w) * S_w, from the top window down  (src/cuzk/msm.rs:411-416)
inline bool combine_windows(const uint8_t* sums96, int num_windows, int window_bits, uint8_t* out) {
  hg1 acc = hg1_identity();
  bool ok = true;
  for (int w = num_windows - 1; w >= 0; w--) {
    for (int k = 0; k < window_bits; k++) acc = hg1_double(acc);
    hg1 s;
    ok &= hg1_from_bytes96(s, sums96 + JB * (size_t)w);
    acc = hg1_add(acc, s);
  }
  hg1_to_bytes96(out, acc);
  return ok;
}

// A window sum from the bucket reduce's bit-plane sums (k_bpr_planes: 16 x 96 B per window -- PR_0 .. PR_7 the row planes, PC_0 .. PC_6 the
// column planes, TC the column total):  S_w = sum_b 2^(b+7) PR_b + sum_b 2^b PC_b + TC,  one Horner chain over the bit positions 14 .. 0.
// 29 group operations at ~0.3 us here instead of ~25 dependent ones at ~7 us each in a narrow GPU tail.
constexpr int PLANES_PER_WINDOW = 16;
inline bool window_sum_from_planes(const uint8_t* planes, hg1& sum) {
  bool ok = true;
  hg1 acc = hg1_identity();
  for (int pos = 14; pos >= 0; pos--) {
    acc = hg1_double(acc);
    hg1 term;
    ok &= hg1_from_bytes96(term, planes + JB * (size_t)(pos >= 7 ? pos - 7 : 8 + pos));
    acc = hg1_add(acc, term);
  }
  hg1 tc;
  ok &= hg1_from_bytes96(tc, planes + JB * (size_t)(PLANES_PER_WINDOW - 1));
  sum = hg1_add(acc, tc);
  return ok;
}
// ... as a 96-byte Jacobian record (one window: the caller spreads the windows of a launch over host threads)
inline bool window_from_planes(const uint8_t* planes, uint8_t* sum96) {
  hg1 s;
  const bool ok = window_sum_from_planes(planes, s);
  hg1_to_bytes96(sum96, s);
  return ok;
}

// The wide fixed-base tables' finish (msm_kernels.h: wide_key).  Magnitude m sits in virtual window vw = (m - 1) mod V at the slot of value
// (m - 1) / V + 1, so with the windows' weighted sums W_vw and plain totals TC_vw (V = nvirt, a power of two):
//     sum_m m B_m = V * sum_vw W_vw - sum_vw (V - 1 - vw) TC_vw = V * sum_vw W_vw - sum_{j=0}^{V-2} (TC_0 + ... + TC_j)
// -- log2 V doublings and 2 additions per virtual window.
// (`sums` / `totals`: W_vw and TC_vw as JB-byte records, `sum_stride` / `total_stride` bytes apart)
inline bool combine_wide_strided(const uint8_t* sums, size_t sum_stride, const uint8_t* totals, size_t total_stride, int nvirt, uint8_t* out) {
  bool ok = true;
  hg1 acc = hg1_identity(), run = hg1_identity(), minus = hg1_identity();
  for (int vw = 0; vw < nvirt; vw++) {
    hg1 w, tc;
    ok &= hg1_from_bytes96(w, sums + sum_stride * (size_t)vw);
    acc = hg1_add(acc, w);
    if (vw < nvirt - 1) {
      ok &= hg1_from_bytes96(tc, totals + total_stride * (size_t)vw);
      run = hg1_add(run, tc);
      minus = hg1_add(minus, run);
    }
  }
  for (int v = nvirt; v > 1; v >>= 1) acc = hg1_double(acc);
  hg1_to_bytes96(out, hg1_add(acc, hg1_neg(minus)));
  return ok;
}
inline bool combine_wide(const uint8_t* sums96, const uint8_t* planes, int nvirt, uint8_t* out) {
  return combine_wide_strided(sums96, JB, planes + JB * (size_t)(PLANES_PER_WINDOW - 1), JB * (size_t)PLANES_PER_WINDOW, nvirt, out);
}
// ... from the (W_hi, TC_hi) pairs the shares of a window-sharded run deliver (k_bpr_final with emit_total): pairs[hi] = W_hi || TC_hi
inline bool combine_wide_pairs(const uint8_t* pairs, int nvirt, uint8_t* out) {
  return combine_wide_strided(pairs, 2 * (size_t)JB, pairs + JB, 2 * (size_t)JB, nvirt, out);
}

}  // namespace host
}  // namespace MSM_FIELD_NS
