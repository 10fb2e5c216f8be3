// Scalar split for the curve endomorphism (SURVEY.md 8f-3 "GLV endomorphism ... to halve scalar length"; the reference uses
// full 254-bit scalars, src/cuzk/msm.rs:79-82).
//
// Both supported curves have j = 0: phi(x, y) = (beta x, y) is the multiplication by lambda, a cube root of unity mod r.  Every
// scalar is written k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^127, so that
//     sum_i k_i P_i  =  sum_i k1_i P_i  +  sum_i k2_i phi(P_i)
// is an MSM over 2n points with 128-bit scalars: the same number of bucket additions, HALF the windows -- half the buckets to
// stitch and reduce, half the window sums to combine on the host.
//
// With a short basis (a1, b1), (a2, b2) of the lattice {(x, y): x + y lambda = 0 mod r} (tools/gen_constants.py derives it,
// det = +r):  c1 = round(k b2 / r), c2 = round(-k b1 / r), (k1, k2) = (k, 0) - c1 (a1, b1) - c2 (a2, b2).
// ANY integers c1, c2 give a valid split (the subtracted vector is in the lattice); the rounding only bounds the size.  Here
//     m = (k * G + 2^319) >> 320,  G = round(2^320 |b| / r)        (|error| < 2^-60: m is the nearest integer or, in a
// 2^-60 neighbourhood of a half, its neighbour), which keeps |k1|, |k2| <= (1/2 + 2^-9) (|a1| + |a2|) < 2^127 - 2^112 (asserted by
// the generator) -- the bound under which the signed 16-bit recode of a 128-bit half cannot carry out of its 8 windows.
// k1 and k2 are computed modulo 2^160 in two's complement (the signs of the basis are folded into the constants N11 .. N22).
//
// Output: magnitude in bits 0 .. 126, sign in bit 127 (4 words).  Works for every k < 2^256, canonical or not.
// Host + device code: tests/host_harness compiles it with g++ against the oracle's model (oracle/bn254_ref.py: glv_split).
#ifndef MSM_CURVE_UNIT
#pragma once
#include "curve_select.h"
#include MSM_CURVE_CONSTANTS
#endif
#include <cstdint>

#if defined(__HIPCC__)
#define GLV_HD __host__ __device__ __forceinline__
#else
#define GLV_HD inline
#endif

namespace MSM_FIELD_NS {

// (k * g + 2^319) >> 320 for k < 2^256 (8 words), g < 2^224 (7 words): 5 words
GLV_HD void glv_mulshift(const uint32_t k[8], const uint32_t g[7], uint32_t m[5]) {
  uint32_t prod[15];
#pragma unroll
  for (int i = 0; i < 15; i++) prod[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const uint64_t t = (uint64_t)k[i] * g[j] + prod[i + j] + carry;
      prod[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    prod[i + 7] = (uint32_t)carry;
  }
  uint64_t c = ((uint64_t)prod[9] + 0x80000000u) >> 32;  // the rounding bit 319 is the top bit of word 9
#pragma unroll
  for (int i = 0; i < 5; i++) {
    c += prod[10 + i];
    m[i] = (uint32_t)c;
    c >>= 32;
  }
}

// acc += m * n  (mod 2^160)
GLV_HD void glv_mac160(uint32_t acc[5], const uint32_t m[5], const uint32_t n[5]) {
#pragma unroll
  for (int i = 0; i < 5; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j + i < 5; j++) {
      const uint64_t t = (uint64_t)m[i] * n[j] + acc[i + j] + carry;
      acc[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
  }
}

// two's complement 160-bit value -> sign (bit 127) and magnitude; false if the magnitude does not fit the recode
GLV_HD bool glv_sign_magnitude(const uint32_t v[5], uint32_t h[4]) {
  const uint32_t neg = v[4] >> 31;
  uint32_t mag[5];
  uint64_t c = neg;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    c += neg ? (uint32_t)~v[i] : v[i];
    mag[i] = (uint32_t)c;
    c >>= 32;
  }
  h[0] = mag[0];
  h[1] = mag[1];
  h[2] = mag[2];
  h[3] = mag[3] | (neg << 31);
  return mag[4] == 0 && mag[3] < 0x7fff0000u;  // < 2^127 - 2^112
}

GLV_HD bool glv_split(const uint32_t k[8], uint32_t h1[4], uint32_t h2[4]) {
  uint32_t m1[5], m2[5];
  glv_mulshift(k, GLV_G1_32, m1);
  glv_mulshift(k, GLV_G2_32, m2);
  uint32_t k1[5] = {k[0], k[1], k[2], k[3], k[4]}, k2[5] = {0, 0, 0, 0, 0};
  glv_mac160(k1, m1, GLV_N11_32);
  glv_mac160(k1, m2, GLV_N12_32);
  glv_mac160(k2, m1, GLV_N21_32);
  glv_mac160(k2, m2, GLV_N22_32);
  const bool ok1 = glv_sign_magnitude(k1, h1);
  const bool ok2 = glv_sign_magnitude(k2, h2);
  return ok1 && ok2;
}

}  // namespace MSM_FIELD_NS
