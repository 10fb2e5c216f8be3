// Device kernels of the cuZK-style BN254 MSM pipeline for gfx950 (MI355X).  Included once by msm_hip.hip.
//
// Pipeline (reference: compute_msm, src/cuzk/msm.rs:75-417) and the HBM layout each stage reads/writes
// (W = windows handled by this GPU, stride = n rounded up to 4):
//
//   bases      u32[n][16]            packed affine, Montgomery (R = 2^261), x || y, 64 B per point, resident
//                                    (+ phi(P_i) as records n .. 2n-1 with the endomorphism, + 15 more tables with fixed-base tables)
//   scalars    u32[n][8]             canonical little-endian (wire format)   (endomorphism: halves u32[2n][4], csrc/glv.h)
//   counts     u32[W][tiles][128]    per-tile coarse-bin histogram, then prefix over tiles
//   tmp_val    u32[W][stride]        point index | sign << 31, in coarse-bin order ; tmp_fine u8[W][stride] = slot & 255
//   val_idxs   u32[W][stride]        point index | sign << 31, grouped by bucket slot
//   col_ptr    u32[W][32769]         start of every bucket slot in val_idxs
//   chunk_slot u32[W][chunks]        bucket slot of the first entry of every SMVP chunk
//   buckets    u32[W][32768][40]     XYZZ records (160 B: 36 limbs + valid flag), Montgomery
//   heads/tails  [W][chunks][40]     partial sums of bucket runs that cross SMVP chunk boundaries
//   rows/cols/parts                  bucket-reduce scratch: 256 row sums, 128 column sums, 3 partial results per window
//   wsums      u8 [W][96]            window sums, Jacobian, canonical little-endian (the only data that leaves the device)
//
// The reference keys its CSC rows by the biased digit (65536 rows per window, transpose.template.wgsl:47-73) and lets
// the SMVP thread visit rows h+k and h-k (smvp.template.wgsl:55-92).  Here the sort key is the bucket slot itself
// (|d| mod 2^15, 32768 rows) and the sign rides in bit 31 of the index, so one bucket is one contiguous run.
#ifndef MSM_CURVE_UNIT
#pragma once
#include "g1.h"
#endif
#include <hip/hip_runtime.h>

#include <utility>

namespace MSM_KERNEL_NS {
using namespace MSM_FIELD_NS;

constexpr int WBITS = 16;   // the reference's window (chunk_size, src/cuzk/msm.rs:79) and the unit of the window-sharding API
constexpr int NWIN = 16;
constexpr int MAXLW = 64;  // local windows one launch may carry: (scalar vectors of the launch) x (windows of each)
constexpr int HALF = 1 << (WBITS - 1);  // 32768 bucket slots per window at 16 bits (the largest window supported)
// Sizes that follow the unit's field (FQ_WORDS packed 32-bit words per coordinate: 8 for the 254 / 255-bit fields, 12 for BLS12-381)
constexpr int CW = FQ_WORDS;        // words of a coordinate on the wire and in the resident bases
constexpr int PT_WORDS = 2 * CW;    // an affine point x || y: 64 B (96 B)
constexpr int JAC_WORDS = 3 * CW;   // a Jacobian record x || y || z: 96 B (144 B)

// Window size as a parameter (the reference hard-codes c, src/cuzk/msm.rs:79-82): C-bit signed digits,
// 2^(C-1) bucket slots per window, NWIN = ceil(255 / C) windows (254-bit scalars + one bit for the recode's carry).
// Small MSMs are dominated by the bucket reduce of 16 x 2^15 mostly empty buckets; a smaller C trades a few more
// additions per point for 16 x / 4 x fewer buckets.  The host picks C from n (msm_hip.hip: pick_window_bits).
// SW = words per scalar the recode reads: 8 (a 254-bit scalar) or 4 (one 127-bit half of the endomorphism split, csrc/glv.h:
// magnitude in bits 0 .. 126, sign in bit 127).
template <int C, int SW = 8>
struct WinCfg {
  static_assert((C >= 10 && C <= 16) || (C >= 17 && C <= 20), "window bits (17 .. 20: the digits of the wide fixed-base tables, k_count_wide)");
  static_assert(SW == 8 || SW == 4, "scalar words");
  static constexpr int BITS = C;
  static constexpr int SBITS = SW == 8 ? 254 : 127;        // bits of the scalar (magnitude)
  static constexpr int NWIN = (SBITS + C) / C;             // 16: 16 | 8, 14: 19 | 10, 12: 22 | 11
  static constexpr int HALF = 1 << (C - 1);                // bucket slots per window
  static constexpr int TBITS = NWIN * C;                   // bits of the biased scalar that carry digits
  static constexpr int WORDS = (TBITS + 31) / 32;          // 8 or 9 | 4 or 5
};
__host__ __device__ constexpr int nwin_of(int bits, bool halves = false) { return ((halves ? 127 : 254) + bits) / bits; }

// (exponent tables live in constant memory; filled from the generated constexpr arrays)
template <int N>
struct cwords {
  uint32_t w[N];
};
template <int N>
constexpr cwords<N> make_cwords(const uint32_t (&src)[N]) {
  cwords<N> r{};
  for (int i = 0; i < N; i++) r.w[i] = src[i];
  return r;
}
// MSM_FQ2 (a G2 unit, csrc/fq2.h: the coordinate field is a quadratic extension): the prime-field point sampler (try-and-increment on x
// with a square root) is not built; a G2 unit samples multiples of the subgroup's generator instead (k_sample_points below).
#ifndef MSM_FQ2
__device__ __constant__ cwords<CW> c_pp1d4 = make_cwords(FQ_PP1D4_32);
#endif

// ------------------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ void ld8(const uint32_t* p, uint32_t w[8]) {
  const uint4 a = reinterpret_cast<const uint4*>(p)[0];
  const uint4 b = reinterpret_cast<const uint4*>(p)[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void st8(uint32_t* p, const uint32_t w[8]) {
  reinterpret_cast<uint4*>(p)[0] = make_uint4(w[0], w[1], w[2], w[3]);
  reinterpret_cast<uint4*>(p)[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// a coordinate's CW packed words (16-byte aligned: CW is a multiple of 4)
__device__ __forceinline__ void ld_coord(const uint32_t* p, uint32_t w[CW]) {
#pragma unroll
  for (int k = 0; k < CW / 4; k++) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[k];
    w[4 * k] = a.x; w[4 * k + 1] = a.y; w[4 * k + 2] = a.z; w[4 * k + 3] = a.w;
  }
}
__device__ __forceinline__ void st_coord(uint32_t* p, const uint32_t w[CW]) {
#pragma unroll
  for (int k = 0; k < CW / 4; k++) reinterpret_cast<uint4*>(p)[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ fq ld_fq(const uint32_t* p) {  // packed -> limbs (no domain change)
  uint32_t w[CW];
  ld_coord(p, w);
  return fq_unpack(w);
}
__device__ __forceinline__ void st_fq(uint32_t* p, const fq& x) {  // x exact, < 2^(32 CW)
  uint32_t w[CW];
  fq_pack(w, x);
  st_coord(p, w);
}
// w >= modulus ?   MOD = 0: Fq modulus p (CW words; in an extension-field unit: ANY of the FQ_EXT components of CW / FQ_EXT words),
// MOD = 1: Fr modulus r (8 words)  (constants fold to immediates)
template <int MOD>
__device__ __forceinline__ bool geq_modulus(const uint32_t* w) {
  constexpr int NW = MOD == 0 ? CW / FQ_EXT : 8;
  bool any = false;
#pragma unroll
  for (int e = 0; e < (MOD == 0 ? FQ_EXT : 1); e++) {
    bool gt = false, lt = false;
#pragma unroll
    for (int i = NW - 1; i >= 0; i--) {
      const uint32_t m = MOD == 0 ? FQ_P32[i] : FR_R32[i];
      gt = gt || (!lt && w[e * NW + i] > m);
      lt = lt || (!gt && w[e * NW + i] < m);
    }
    any = any || !lt;
  }
  return any;
}
__device__ __forceinline__ bool fq_equal_exact(const fq& a, const fq& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
__device__ __forceinline__ fq fq_curve_b() {  // the curve constant b (y^2 = x^3 + b), Montgomery form
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = FQ_B29[i];
  return r;
}

// Jacobian record, canonical non-Montgomery integers (the wire format of results)
__device__ __forceinline__ void st_jacobian_plain(uint32_t* p, const g1_xyzz& a) {
  fq X, Y, Z;
  g1_to_jacobian(a, X, Y, Z);
  st_fq(p, fq_from_mont(X));
  st_fq(p + CW, fq_from_mont(Y));
  st_fq(p + 2 * CW, fq_from_mont(Z));
}
__device__ __forceinline__ g1_xyzz ld_jacobian_plain(const uint32_t* p) {
  const fq X = fq_to_mont(ld_fq(p)), Y = fq_to_mont(ld_fq(p + CW)), Z = fq_to_mont(ld_fq(p + 2 * CW));
  return g1_from_jacobian(X, Y, Z);
}

// XYZZ record in scratch memory / LDS: 4 FQ_L limbs + identity flag (37 words with 9 limbs: an odd stride, no LDS bank conflicts)
constexpr int XYZZ_WORDS = 4 * FQ_L + 1;
template <typename PTR>
__device__ __forceinline__ void st_xyzz(PTR p, const g1_xyzz& a) {
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    p[i] = a.x.v[i];
    p[FQ_L + i] = a.y.v[i];
    p[2 * FQ_L + i] = a.zz.v[i];
    p[3 * FQ_L + i] = a.zzz.v[i];
  }
  p[4 * FQ_L] = a.inf ? 1u : 0u;
}
template <typename PTR>
__device__ __forceinline__ g1_xyzz ld_xyzz(PTR p) {
  g1_xyzz a;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) {
    a.x.v[i] = p[i];
    a.y.v[i] = p[FQ_L + i];
    a.zz.v[i] = p[2 * FQ_L + i];
    a.zzz.v[i] = p[3 * FQ_L + i];
  }
  a.inf = p[4 * FQ_L] != 0;
  return a;
}

// error bits written to the context's device error word
constexpr uint32_t ERRBIT_NONCANONICAL = 1u;
constexpr uint32_t ERRBIT_NOT_ON_CURVE = 2u;
constexpr uint32_t ERRBIT_SCALAR_CARRY = 4u;
constexpr uint32_t INFOBIT_HUGE_BIN = 0x100u;  // not an error: the fine sort met a coarse bin beyond FINE_BIG (skewed scalars, or large n) -- the host's cue to run k_fine_hist

// ------------------------------------------------------------------------------------------------ stage 0: bases
// canonical wire bytes -> packed Montgomery affine (≙ decompose_scalars.template.wgsl:41-70, the point half)
__global__ void __launch_bounds__(256) k_convert_points(const uint32_t* in, uint32_t* out, size_t n,  // in may alias out (element-wise)
                                                        uint32_t flags, uint32_t* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[CW], wy[CW];
  ld_coord(in + i * PT_WORDS, wx);
  ld_coord(in + i * PT_WORDS + CW, wy);
  if (geq_modulus<0>(wx) || geq_modulus<0>(wy)) atomicOr(err, ERRBIT_NONCANONICAL);
  // flags bit 1 (MSM_HIP_BASES_MONT256): the words are x * 2^256 mod p, not x
  const bool m256 = (flags & 2u) != 0;
  const fq x = m256 ? fq_from_mont256(fq_unpack(wx)) : fq_to_mont(fq_unpack(wx));
  const fq y = m256 ? fq_from_mont256(fq_unpack(wy)) : fq_to_mont(fq_unpack(wy));
  if (flags & 1u) {
    const fq lhs = fq_canonical(fq_sqr(y));
    const fq rhs = fq_canonical(fq_tidy(fq_add(fq_mul(fq_sqr(x), x), fq_curve_b())));
    if (!fq_equal_exact(lhs, rhs)) atomicOr(err, ERRBIT_NOT_ON_CURVE);
  }
  st_fq(out + i * PT_WORDS, x);
  st_fq(out + i * PT_WORDS + CW, y);
}

// The endomorphism's point half (csrc/glv.h): record n + i = phi(P_i) = (beta x_i, y_i) behind the n plain bases
// (a G2 unit: beta is the element (beta, 0) of Fq2 -- the twist has j = 0 like the curve, tools/gen_constants.py emit_g2)
// (records first .. first + count - 1 of the n: the one-shot entry point converts the bases chunk by chunk as they arrive)
__global__ void __launch_bounds__(256) k_endo_points(uint32_t* __restrict__ bases, size_t n, size_t first, size_t count) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const size_t i = first + k;
  fq beta;
#pragma unroll
  for (int k = 0; k < FQ_L; k++) beta.v[k] = FQ_BETA29[k];
  st_fq(bases + (n + i) * PT_WORDS, fq_mul(ld_fq(bases + i * PT_WORDS), beta));
  const uint4* y = reinterpret_cast<const uint4*>(bases + i * PT_WORDS + CW);
  uint4* o = reinterpret_cast<uint4*>(bases + (n + i) * PT_WORDS + CW);
#pragma unroll
  for (int k = 0; k < CW / 4; k++) o[k] = y[k];
}

// Fixed-base tables (reference README.md "Future work": the Elastic-MSM precomputation trade-off): with
// T_w[i] = 2^(16 w) P_i stored for every window, sum_i s_i P_i = sum_i sum_w d_{i,w} T_w[i] needs ONE bucket set for all
// windows -- one stitch / bucket reduce instead of 16 and no window combine -- for 16 x the base memory.
// bases[(w * nb + i)][16]: table w behind table w - 1; table 0 is the plain converted base set (so every entry point that does
// not use the tables keeps working on the same buffer).  One thread per point: 16 doublings per table in XYZZ, then back to
// affine (one inversion by Fermat, a^(p-2)).
__device__ __constant__ cwords<CW / FQ_EXT> c_pm2 = make_cwords(FQ_PM2_32);  // p - 2 (the prime field's)
#ifndef MSM_FQ2
__device__ __forceinline__ fq fq_inv(const fq& a) {  // a exact, nonzero; result exact, < 2p
  fq acc = fq_one();
  for (int bit = 32 * CW - 1; bit >= 0; bit--) {  // (leading zero bits of p - 2 only square the initial one)
    acc = fq_sqr(acc);
    if ((c_pm2.w[bit >> 5] >> (bit & 31)) & 1u) acc = fq_mul(acc, a);
  }
  return acc;
}
#else
__device__ __forceinline__ fq fq_inv(const fq& a) {  // Fq2: 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2), one Fermat inversion in the prime field
  const fp a0 = f2_c0(a), a1 = f2_c1(a);
  const fp nrm = fpn::fq_mul2(a0, a0, a1, a1);  // the norm, exact, < 2p
  fp acc = fpn::fq_one();
  for (int bit = 32 * FP_WORDS - 1; bit >= 0; bit--) {
    acc = fpn::fq_sqr(acc);
    if ((c_pm2.w[bit >> 5] >> (bit & 31)) & 1u) acc = fpn::fq_mul(acc, nrm);
  }
  return f2_make(fpn::fq_mul(a0, acc), fpn::fq_mul(fpn::fq_sub<3>(fpn::fq_zero(), a1), acc));  // (3p - a1) / norm
}
#endif
// (step_bits: doublings between two tables -- 16, or 20 for the wide tables below, whose last table is only last_step_bits above the one before)
__global__ void __launch_bounds__(256) k_precompute_tables(uint32_t* __restrict__ bases, size_t n, size_t nb, int num_tables, int step_bits,
                                                           int last_step_bits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fq px = ld_fq(bases + i * PT_WORDS), py = ld_fq(bases + i * PT_WORDS + CW);
  g1_xyzz acc = g1_from_affine(px, py);
  for (int w = 1; w < num_tables; w++) {
    const int steps = w == num_tables - 1 ? last_step_bits : step_bits;
#pragma unroll 1
    for (int k = 0; k < steps; k++) acc = g1_double(acc);
    // affine again: x = X / ZZ, y = Y / ZZZ with one inversion of ZZ * ZZZ (a point of prime order never doubles to infinity)
    const fq t = fq_inv(fq_mul(acc.zz, acc.zzz));
    const fq x = fq_canonical(fq_mul(acc.x, fq_mul(t, acc.zzz)));
    const fq y = fq_canonical(fq_mul(acc.y, fq_mul(t, acc.zz)));
    st_fq(bases + ((size_t)w * nb + i) * PT_WORDS, x);
    st_fq(bases + ((size_t)w * nb + i) * PT_WORDS + CW, y);
    acc = g1_from_affine(x, y);
  }
}

// ------------------------------------------------------------------------------------------------ stage 1+2: recode + sort
// Signed 16-bit digit recode (≙ decompose_scalars.template.wgsl:83-112, CPU model test/utils.rs:121-161):
//   d = raw + carry; if d >= 2^15 { d -= 2^16; carry = 1 }  -- computed per window without the serial carry chain.  Signed-magnitude code = sign << 15 | (|d| & 0x7fff):
//   0 = digit 0 (contributes nothing), 0x8000 = digit -2^15 (bucket slot 0).
//
// The reference's transpose (transpose.template.wgsl:32-76) is a counting sort run by 16 threads.  Here it is a
// two-level LDS counting sort over the 15-bit bucket slot, and the recode is fused into both of its global passes
// (scalars are re-read instead of materialising 16 digit planes: 32 B per scalar either way):
//   k_count          per tile of scalars: LDS histogram of the 128 coarse bins (slot >> 8) of every window
//   k_scan_tiles     per (window, coarse bin): prefix over tiles, bin totals
//   k_scatter_coarse per tile: LDS-ranked scatter of (index | sign << 31, slot & 255) into coarse-bin order
//   k_sort_fine      per (window, coarse bin): LDS counting sort over its 256 slots -> val_idxs + col_ptr
// Order inside a slot is the arrival order of LDS atomics; the group sum does not depend on it.
constexpr int NCOARSE = 128;       // coarse bins per window
constexpr int FINE = HALF / NCOARSE;  // 256 slots per coarse bin

// Adding 0x8000 to every 16-bit halfword of the 256-bit scalar (one multiword addition) performs the whole carry chain
// at once: halfword w of t = s + 0x8000...8000 is the reference's biased digit d_w + 2^15 (decompose_scalars.template.wgsl:
// 105-112), and the carry out of bit 255 is its "final carry".  Each window's digit is then read independently.
// The same for C-bit windows: the bias constant has bit C w + C - 1 set for every window w (word i of it below), the biased
// scalar t has WinCfg<C>::WORDS words, and the recode overflows iff t has a bit at or above C * NWIN.
template <int C, int SW = 8>
__host__ __device__ constexpr uint32_t bias_word(int i) {
  uint32_t v = 0;
  for (int w = 0; w < WinCfg<C, SW>::NWIN; w++) {
    const int bit = C * w + C - 1;
    if (bit / 32 == i) v |= 1u << (bit % 32);
  }
  return v;
}
template <int C, int SW = 8>
__device__ __forceinline__ uint32_t bias_scalar(const uint32_t s[SW], uint32_t t[WinCfg<C, SW>::WORDS]) {
  constexpr int WORDS = WinCfg<C, SW>::WORDS;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < WORDS; i++) {
    c += (uint64_t)(i < SW ? s[i] : 0u) + bias_word<C, SW>(i);
    t[i] = (uint32_t)c;
    c >>= 32;
  }
  // 1: the recode does not fit NWIN windows ("final carry is 1", test/utils.rs:150-152)
  if constexpr (WinCfg<C, SW>::TBITS == 32 * WORDS) return (uint32_t)c;
  else return (t[WORDS - 1] >> (WinCfg<C, SW>::TBITS - 32 * (WORDS - 1))) != 0u ? 1u : 0u;
}
// the recode's input: a scalar (8 words) or one half of the endomorphism split (4 words; `neg` receives its sign)
template <int SW>
__device__ __forceinline__ void ld_scalar(const uint32_t* p, uint32_t s[SW], uint32_t& neg) {
  if constexpr (SW == 8) {
    ld8(p, s);
    neg = 0;
  } else {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    s[0] = a.x; s[1] = a.y; s[2] = a.z;
    s[3] = a.w & 0x7fffffffu;
    neg = a.w >> 31;
  }
}
// biased digit b = d + 2^(C-1) of window w  ->  signed-magnitude code: sign << 15 | (|d| mod 2^(C-1))
template <int C>
__device__ __forceinline__ uint32_t code_of_window(const uint32_t* t, int w) {  // t: WinCfg<C, SW>::WORDS words
  constexpr uint32_t H = (uint32_t)WinCfg<C>::HALF;
  const int bit = C * w, i = bit >> 5, sh = bit & 31;
  uint32_t b = t[i] >> sh;
  if (sh + C > 32) b |= t[i + 1] << (32 - sh);  // (only then is i + 1 < WORDS)
  b &= (1u << C) - 1u;
  if (b >= H) return b - H;                       // d = 0 .. 2^(C-1) - 1 (0: no entry)
  return 0x8000u | ((H - b) & (H - 1u));          // d = -(2^(C-1) - b): magnitude 1 .. 2^(C-1) (2^(C-1) -> slot 0)
}

// Scalars handed over as s * 2^256 mod r (the in-memory limbs of a 4 x 64-bit Montgomery library with R = 2^256) are turned
// into the canonical wire format by one pre-pass: a 9-limb Montgomery reduction of (s_mont << 5), i.e. s_mont * 2^5 / 2^261.
#ifdef MSM_FQ2  // (a G2 unit: the container of a 256-bit scalar is an element of the PRIME field)
using sfe = fp;
constexpr int SF_L = FP_L, SF_WORDS = FP_WORDS;
__device__ __forceinline__ sfe sf_unpack(const uint32_t* w) { return fpn::fq_unpack(w); }
__device__ __forceinline__ void sf_pack(uint32_t* w, const sfe& x) { fpn::fq_pack(w, x); }
#else
using sfe = fq;
constexpr int SF_L = FQ_L, SF_WORDS = CW;
__device__ __forceinline__ sfe sf_unpack(const uint32_t* w) { return fq_unpack(w); }
__device__ __forceinline__ void sf_pack(uint32_t* w, const sfe& x) { fq_pack(w, x); }
#endif
__device__ __forceinline__ void fr_from_mont256(const uint32_t w[8], uint32_t out[8]) {
  // (the scalar field's 256-bit values in the unit's limb layout: SF_L limbs of FQ_W bits hold them with room to spare)
  constexpr int SH = FQ_W * SF_L - 256, SH_LIMBS = SH / FQ_W, SH_BITS = SH % FQ_W;  // s_mont * 2^SH / 2^(FQ_W SF_L) = s_mont / 2^256
  uint32_t wide[SF_WORDS];
#pragma unroll
  for (int k = 0; k < SF_WORDS; k++) wide[k] = k < 8 ? w[k] : 0u;
  const sfe x = sf_unpack(wide);
  uint64_t c[2 * SF_L + 1];
#pragma unroll
  for (int k = 0; k < 2 * SF_L + 1; k++) c[k] = 0;
#pragma unroll
  for (int k = 0; k < SF_L; k++) c[k + SH_LIMBS] = (uint64_t)x.v[k] << SH_BITS;
#pragma unroll
  for (int i = 0; i < SF_L; i++) {
    const uint32_t m = ((uint32_t)c[i] * FR_N0_29) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < SF_L; j++) c[i + j] += (uint64_t)m * FR_R29[j];
    c[i + 1] += c[i] >> FQ_W;
  }
  sfe t;
#pragma unroll
  for (int k = SF_L; k < 2 * SF_L - 1; k++) {
    t.v[k - SF_L] = (uint32_t)c[k] & FQ_MASK;
    c[k + 1] += c[k] >> FQ_W;
  }
  t.v[SF_L - 1] = (uint32_t)c[2 * SF_L - 1];
  // t <= r: one conditional subtraction makes it canonical
  sfe d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < SF_L; i++) {
    const uint32_t u = t.v[i] - FR_R29[i] - borrow;
    borrow = u >> 31;
    d.v[i] = (i < SF_L - 1) ? (u & FQ_MASK) : u;
  }
#pragma unroll
  for (int i = 0; i < SF_L; i++) t.v[i] = borrow ? t.v[i] : d.v[i];
  sf_pack(wide, t);
#pragma unroll
  for (int k = 0; k < 8; k++) out[k] = wide[k];
}

__global__ void __launch_bounds__(256) k_scalars_from_mont256(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t count,
                                                              uint32_t* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t w[8], o[8];
  ld8(in + i * 8, w);
  if (geq_modulus<1>(w)) atomicOr(err, ERRBIT_NONCANONICAL);
  fr_from_mont256(w, o);
  uint4* q = reinterpret_cast<uint4*>(out + i * 8);
  q[0] = make_uint4(o[0], o[1], o[2], o[3]);
  q[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// `nvec` scalar vectors (vec_stride words apart) may share one launch: vector v, window w is handled as local window
// lw = v * w_count + (w - w_begin), nvec * w_count <= MAXLW -- several MSMs over the same bases sorted, accumulated and reduced
// by one kernel sequence (used by the window-sharded multi-GPU pipeline, where one MSM's share is too small to fill a GPU).
//
// Digit planes.  `planes` receives every local window's digit code, u16 planes[lw][n] (PLANE_MODE below): for the debug read-back
// (msm_hip_read_digits), or as the input of the launch's second pass (k_scatter_planes reads them instead of the scalars).
// A rank of a window-sharded run needs 1 - 4 of a scalar's 16 digits: its second pass then reads 2 - 8 B per scalar instead of 32,
// and keeps no scalar in registers (the scalar-reading scatter holds 8 biased scalars per thread: 282 VGPRs at 16 bits).
// PLANE_MODE of k_count's `planes` output: 0 none; 1 debug read-back (the half's sign folded into bit 15); 2 raw codes for k_scatter_planes
// (with the signs of the halves, if any, in `negbits`).
// SPLIT (endomorphism launches, SW = 4): `scalars` are the nvec x n / 2 full 8-word scalars; the kernel splits each into its two halves
// (csrc/glv.h) itself -- the separate pass of round 2 (k_glv_split: 32 B read + 32 B written per scalar and a kernel of its own in front of
// every launch) is gone -- and treats them as inputs 2 j (k1, multiplies P_j) and 2 j + 1 (k2, multiplies phi(P_j)) of the 2n-input problem:
// INTERLEAVED positions, so that a tile of positions is a tile of scalars and one LDS histogram serves both halves.  The halves go to
// `halves_out` (position p at word 4 p: the same 32 B the scalar took) for k_scatter_coarse<C, 4>; negbits[v][h][n / 128 rounded up]: bit j of
// half h's array is the sign of half h of scalar j.
template <int C, int SW, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_count(const uint32_t* __restrict__ scalars, size_t n, uint32_t tile_len, uint32_t tiles,
                                               int w_begin, int w_count, int nvec, size_t vec_stride,
                                               uint32_t* __restrict__ counts, uint16_t* __restrict__ planes, int plane_mode,
                                               uint64_t* __restrict__ negbits, uint32_t* __restrict__ halves_out,
                                               uint32_t* __restrict__ err, size_t merge_nb) {
  static_assert(!SPLIT || SW == 4, "the split produces 4-word halves");
  // merge_nb != 0 (fixed-base tables, see k_precompute_tables): every window of vector v feeds ONE bucket set, local window v
  // grid (tiles, nvec): a workgroup counts one tile of ONE scalar vector (round 4: with the vectors looped over inside the workgroup a
  // grouped launch of small MSMs kept a quarter of the CUs busy -- 64 tiles at 2^16 -- for nvec times as long)
  __shared__ uint32_t cnt[MAXLW * NCOARSE];
  const int tid = threadIdx.x;
  const int v = blockIdx.y;
  const int le0 = merge_nb ? v : v * w_count, le_n = merge_nb ? 1 : w_count;  // this vector's local windows
  (void)nvec;
  for (int i = tid; i < le_n * NCOARSE; i += 256) cnt[le0 * NCOARSE + i] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * tile_len;
  const size_t end = base + tile_len < n ? base + tile_len : n;
  uint32_t bad = 0;
  // one recoded input: histogram and plane entries of its local windows
  auto emit = [&](int v, size_t pos, const uint32_t* tb, uint32_t neg) {
#pragma unroll
    for (int w = 0; w < WinCfg<C, SW>::NWIN; w++) {
      const int lw = w - w_begin;
      if (lw >= 0 && lw < w_count) {
        const int le = merge_nb ? v : v * w_count + lw;
        const uint32_t code = code_of_window<C>(tb, w);
        if (code != 0) atomicAdd(&cnt[le * NCOARSE + ((code & 0x7fffu) >> 8)], 1u);
        if (plane_mode) planes[((size_t)v * w_count + lw) * n + pos] = (uint16_t)(plane_mode == 2 ? code : (code ? code ^ (neg << 15) : 0u));
      }
    }
  };
  {
    const uint32_t* sv = scalars + (size_t)v * vec_stride;
    if constexpr (SPLIT) {
      const size_t nsc = n / 2, neg_words = (nsc + 63) / 64;
      for (size_t j0 = base / 2; j0 < end / 2; j0 += 256) {  // (tile_len is a multiple of 256 positions: a wave's 64 scalars share a word of negbits)
        const size_t j = j0 + tid;
        const bool valid = j < end / 2;
        uint32_t k[8], h[2][4];
#pragma unroll
        for (int q = 0; q < 8; q++) k[q] = 0;
        if (valid) ld8(sv + j * 8, k);
        // the input contract of the plain path: scalars that overflow the reference's 16-bit recode are rejected (test/utils.rs:150-152)
        uint64_t c = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) c = (c + k[q] + 0x80008000u) >> 32;
        const bool ok = glv_split(k, h[0], h[1]);
        if (c != 0 || !ok) bad = 1;
        if (negbits) {
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            const unsigned long long nb = __ballot((h[hh][3] >> 31) != 0u);
            if ((tid & 63) == 0 && valid) negbits[((size_t)v * 2 + hh) * neg_words + j / 64] = nb;
          }
        }
        if (!valid) continue;
        if (halves_out) {
          uint4* o = reinterpret_cast<uint4*>(halves_out + ((size_t)v * nsc + j) * 8);
          o[0] = make_uint4(h[0][0], h[0][1], h[0][2], h[0][3]);
          o[1] = make_uint4(h[1][0], h[1][1], h[1][2], h[1][3]);
        }
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          uint32_t s[4] = {h[hh][0], h[hh][1], h[hh][2], h[hh][3] & 0x7fffffffu}, tb[WinCfg<C, 4>::WORDS];
          bad |= bias_scalar<C, 4>(s, tb);
          emit(v, 2 * j + hh, tb, h[hh][3] >> 31);
        }
      }
    } else {
      for (size_t i0 = base; i0 < end; i0 += 256) {
        const size_t i = i0 + tid;
        if (i >= end) continue;
        uint32_t s[SW], tb[WinCfg<C, SW>::WORDS], neg = 0;
        ld_scalar<SW>(sv + i * SW, s, neg);
        bad |= bias_scalar<C, SW>(s, tb);
        if constexpr (C != 16 && SW == 8) {  // the same input contract for every window size: scalars that overflow the reference's
          uint32_t t16[8];                   // 16-bit recode ("final carry is 1", test/utils.rs:150-152) are rejected
          bad |= bias_scalar<16>(s, t16);
        }
        emit(v, i, tb, neg);
      }
    }
  }
  if (bad) atomicOr(err, ERRBIT_SCALAR_CARRY);
  __syncthreads();
  // counts[lw][tile][bin]
  for (int i = tid; i < le_n * NCOARSE; i += 256)
    counts[((size_t)(le0 + i / NCOARSE) * tiles + blockIdx.x) * NCOARSE + (i % NCOARSE)] = cnt[le0 * NCOARSE + i];
}

// One wave per (window, coarse bin): in place, counts[lw][tile][bin] becomes the number of entries of that bin in earlier
// tiles; bin_total[lw][bin] receives the bin's size.  (The 128 totals of a window are turned into bin starts by every
// workgroup of k_scatter_coarse for itself: a last-block hand-off here needs agent-scope releases, i.e. L2 write-backs,
// which cost 70 us.)
__global__ void __launch_bounds__(256) k_scan_tiles(uint32_t* __restrict__ counts, uint32_t tiles, uint32_t* __restrict__ bin_total) {
  const int lw = blockIdx.y, lane = threadIdx.x & 63;
  const int bin = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t* c = counts + (size_t)lw * tiles * NCOARSE + bin;
  uint32_t run = 0;
  for (uint32_t t0 = 0; t0 < tiles; t0 += 64) {
    const uint32_t t = t0 + lane;
    const uint32_t v = t < tiles ? c[(size_t)t * NCOARSE] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (t < tiles) c[(size_t)t * NCOARSE] = run + x - v;
    run += __shfl(x, 63);
  }
  if (lane == 0) bin_total[lw * NCOARSE + bin] = run;
}

// Exclusive prefix sum of one value per thread over a 256-thread block (4 waves); `wave_tot` is 4 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* wave_tot) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_tot[wid] = x;
  __syncthreads();
  uint32_t add = 0;
  for (int k = 0; k < wid; k++) add += wave_tot[k];
  __syncthreads();
  return x - v + add;
}

// The SMVP's chunk length is chosen on the DEVICE from the number of entries the sort actually produced: the host sizes the chunk
// arrays and grids for n entries per window (`chunks` lanes of `host_len` entries), but zero digits produce no entry -- witness-like
// scalar vectors (many zeros and ones) fill a fraction of that, and with the host's length most lanes would find nothing to do while
// the rest carry full-length chunks.  Every kernel that uses the chunk structure (k_sort_fine's chunk table, k_smvp_chunks, the
// stitch kernels) uses the same length: the largest window's entries spread over all `chunks` lanes.  It is computed ONCE per launch, by
// workgroup 0 of k_scatter_coarse (which scans the windows' bin totals anyway), into a word of the launch's slot (`chunk_len_dev`);
// the consumers load that one word (deriving it per workgroup from the 16 .. 64 window totals cost every SMVP workgroup a chain of
// scalar loads at its start: +1 % on the whole MSM).
constexpr int SMVP_CHUNK_MIN_ENTRIES = 8;
// (rounds 1 - 2 kept chunk lengths multiples of 4 for the index loads of that time; any length works since the SMVP loads one index per entry,
//  and the host now picks the length by the workgroups-per-CU count it produces: msm_hip.hip, chunk_len_for)
constexpr uint32_t SMVP_CHUNK_ROUND = 1;
__device__ __forceinline__ uint32_t smvp_chunk_len(uint32_t mx, uint32_t chunks, uint32_t host_len) {
  uint32_t len = (uint32_t)(((uint64_t)mx + chunks - 1) / chunks);
  len = (len + (SMVP_CHUNK_ROUND - 1u)) / SMVP_CHUNK_ROUND * SMVP_CHUNK_ROUND;
  if (len < (uint32_t)SMVP_CHUNK_MIN_ENTRIES) len = SMVP_CHUNK_MIN_ENTRIES;
  return len < host_len ? len : host_len;
}
// Both scatter kernels stage their output through LDS: the block ranks its items per destination bin with LDS atomics,
// lays them out bin-major in LDS, and writes them out in LDS order, so consecutive lanes store to consecutive global
// addresses inside each (tile, bin) run instead of 64 unrelated 4-byte stores per wave instruction.
constexpr int SCAT_SUB = 2048;  // scalars staged per block iteration (8 per thread)

// (round 5) The run cursors (gpos) live in dynamic LDS, as many as the launch has local windows (512 B each): with the 32 KB of a 64-window launch
// declared statically, three workgroups fitted a CU whatever the launch's size; the half-scalar form is held to 128 registers (four waves per SIMD):
// 1290 -> 1148 us at 2^24.  (Tried and dropped: splitting the scalars again here instead of reading the halves the first pass wrote -- 1 GB less
// traffic at 2^24, and 1522 us instead of 1148 with the first pass no faster: profiles/r05_sort.txt.)
template <int C, int SW>
__global__ void __launch_bounds__(256, (SW == 4 ? 4 : 1)) k_scatter_coarse(const uint32_t* __restrict__ scalars, size_t n, size_t stride, uint32_t tile_len,
                                                        uint32_t tiles, int w_begin, int w_count, int nvec, size_t vec_stride,
                                                        const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ bin_total, uint32_t* __restrict__ coarse_ptr,
                                                        uint32_t* __restrict__ tmp_val,
                                                        uint8_t* __restrict__ tmp_fine, size_t merge_nb, uint32_t half_n, uint32_t half_shift,
                                                        uint32_t chunks, uint32_t host_chunk_len, uint32_t* __restrict__ chunk_len_dev) {
  // scalars a thread holds (biased, in registers) per block iteration: 8 halves of 4 words, or 4 full scalars of 8 words -- 8 of those cost
  // 282 VGPRs + 26 AGPRs at 16 bits (one wave per SIMD) and a 304-byte scratch object at 12 bits (round 3)
  constexpr int PER = SW == 8 ? 4 : 8;
  constexpr int SUB = 256 * PER;
  static_assert(SUB <= SCAT_SUB, "LDS staging arrays");
  // SW = 4 (endomorphism halves, interleaved by k_count<C, 4, true>): input 2 j is k1 of scalar j and multiplies base j; input 2 j + 1 is
  // k2 and multiplies phi(P_j), record half_shift = n_bases + j
  extern __shared__ uint32_t gpos[];  // [local windows of the launch][NCOARSE]: global write cursor of every (window, coarse bin) run of this tile
  __shared__ uint32_t hist[NCOARSE];
  __shared__ uint32_t lstart[NCOARSE];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t st_val[SCAT_SUB];
  __shared__ uint32_t st_dst[SCAT_SUB];
  __shared__ uint8_t st_fine[SCAT_SUB];
  __shared__ uint32_t max_total;  // entries of the fullest local window (workgroup 0: -> chunk_len_dev)
  const int tid = threadIdx.x;
  if (tid == 0) max_total = 0;
  __syncthreads();
  // start of every (window, coarse bin): exclusive scan of the window's 128 bin totals -- a pair of waves per window, two
  // windows per step; workgroup 0 also publishes them as coarse_ptr[lw][0..128] for k_sort_fine
  // grid (tiles, nvec): a workgroup scatters one tile of ONE scalar vector and needs the starts of that vector's windows only; workgroup (0, 0)
  // scans the windows of every vector: it publishes all of them and the launch's chunk length
  const int w_eff = merge_nb ? nvec : nvec * w_count;  // local windows of all vectors of this launch
  const int v = blockIdx.y;
  const bool publisher = blockIdx.x == 0 && blockIdx.y == 0;
  const int le0 = publisher ? 0 : (merge_nb ? v : v * w_count), le1 = publisher ? w_eff : le0 + (merge_nb ? 1 : w_count);
  for (int i0 = le0 * NCOARSE; i0 < le1 * NCOARSE; i0 += 256) {
    const int i = i0 + tid, lw = i / NCOARSE, bin = i % NCOARSE, lane = tid & 63;
    const bool live = i < le1 * NCOARSE;  // odd window counts: the last step has one idle pair of waves
    const uint32_t bt = live ? bin_total[i] : 0u;
    uint32_t x = bt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[tid >> 6] = x;
    __syncthreads();
    const uint32_t incl = x + ((tid >> 6) & 1 ? wave_tot[(tid >> 6) - 1] : 0u);
    if (live) gpos[i] = incl - bt + counts[((size_t)lw * tiles + blockIdx.x) * NCOARSE + bin];
    if (live && publisher) {
      coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin] = incl - bt;
      if (bin == NCOARSE - 1) {
        coarse_ptr[(size_t)lw * (NCOARSE + 1) + NCOARSE] = incl;
        atomicMax(&max_total, incl);
      }
    }
    __syncthreads();
  }
  if (publisher && tid == 0) *chunk_len_dev = smvp_chunk_len(max_total, chunks, host_chunk_len);
  const size_t tile_base = (size_t)blockIdx.x * tile_len;
  const size_t tile_end = tile_base + tile_len < n ? tile_base + tile_len : n;
  for (size_t sub = tile_base; sub < tile_end; sub += SUB) {
    // this thread's PER biased scalars stay in registers; every window's digit code is read from them
    const uint32_t* sv = scalars + (size_t)v * vec_stride;
    uint32_t sc[PER][WinCfg<C, SW>::WORDS];
    uint32_t negs = 0;  // bit j: scalar j is a negative half (its digits' signs are flipped)
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const size_t i = sub + (size_t)j * 256 + tid;
      uint32_t raw[SW], neg = 0;
#pragma unroll
      for (int k = 0; k < SW; k++) raw[k] = 0;  // an all-zero scalar recodes to all-zero digits: no entries
      if (i < tile_end) ld_scalar<SW>(sv + i * SW, raw, neg);
      negs |= neg << j;
      (void)bias_scalar<C, SW>(raw, sc[j]);
    }
#pragma unroll
    for (int w = 0; w < WinCfg<C, SW>::NWIN; w++) {
      if (w < w_begin || w >= w_begin + w_count) continue;  // block-uniform
      // fixed-base tables: window w of point i is table entry w * merge_nb + i, and all windows share local window v
      const int lw = merge_nb ? v : v * w_count + (w - w_begin);
      const uint32_t idx_base = merge_nb ? (uint32_t)(w * merge_nb) : 0u;
      if (tid < NCOARSE) hist[tid] = 0;
      __syncthreads();
      uint32_t rank[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const uint32_t code = code_of_window<C>(sc[j], w);
        rank[j] = code ? atomicAdd(&hist[(code & 0x7fffu) >> 8], 1u) : 0u;
      }
      __syncthreads();
      const uint32_t mine = tid < NCOARSE ? hist[tid] : 0u;
      const uint32_t excl = block_excl_scan_256(mine, wave_tot);
      if (tid < NCOARSE) lstart[tid] = excl;
      __syncthreads();
      const uint32_t total = lstart[NCOARSE - 1] + hist[NCOARSE - 1];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const uint32_t code = code_of_window<C>(sc[j], w);
        if (code) {
          const uint32_t slot = code & 0x7fffu, bin = slot >> 8;
          const uint32_t e = lstart[bin] + rank[j];
          uint32_t pos = (uint32_t)(sub + (size_t)j * 256 + tid);
          if constexpr (SW == 4) pos = (pos >> 1) + ((pos & 1u) ? half_shift : 0u);
          st_val[e] = (idx_base + pos) | (((code >> 15) ^ ((negs >> j) & 1u)) << 31);
          st_fine[e] = (uint8_t)(slot & 0xffu);
          st_dst[e] = gpos[lw * NCOARSE + bin] + rank[j];
        }
      }
      __syncthreads();
      uint32_t* ov = tmp_val + (size_t)lw * stride;
      uint8_t* of = tmp_fine + (size_t)lw * stride;
      for (uint32_t e = tid; e < total; e += 256) {
        const uint32_t d = st_dst[e];
        ov[d] = st_val[e];
        of[d] = st_fine[e];
      }
      if (tid < NCOARSE) gpos[lw * NCOARSE + tid] += hist[tid];
      __syncthreads();
    }
  }
}

// ---- wide fixed-base tables (round 4; SURVEY.md 8f-2, MSM_HIP_BASES_PRECOMPUTE_WIDE) ------------------------------------------------
// With tables T_w[i] = 2^(C w) P_i the number of bucket additions of an MSM is ceil(255 / C) * n and nothing ties C to the bucket count of
// a window any more -- there is one bucket set of 2^(C-1) slots.  C = 17 / 19 / 20: 15 / 14 / 13 additions per point instead of 16 (the SMVP,
// the dominant kernel, shrinks by that much).  The slot of magnitude m (1 .. 2^(C-1)) is split as
//     m = hi * 2^15 + value(slot),   hi = (m - 1) >> 15,   slot = m & 0x7fff,   value(slot) = slot, or 2^15 for slot 0
// and `hi` is handled as a VIRTUAL WINDOW: local window hi holds the 2^15 slots of that range, so that everything behind the two
// scalar-reading passes -- fine sort, SMVP, stitch, row / column sums -- runs unchanged on 2^(C-16) local windows of 2^15 slots.  The reduce leaves,
// per virtual window, the weighted sum W_hi = sum_slot value(slot) B[hi][slot] AND the plain total TC_hi = sum_slot B[hi][slot] (the column
// total of the bit-plane sums, k_bpr_planes), and the host finishes  sum_hi W_hi + 2^15 * sum_hi hi * TC_hi  (host_g1.h: combine_wide).
// The entries of virtual window hi are stored at tmp_val[hi][...]: with skewed scalars one virtual window may receive all T n entries, so
// the per-window stride is T n (the host sizes the arrays for it); the lanes of the SMVP are sized for the uniform case and the device
// picks the chunk length from the fullest window as always (smvp_chunk_len).
// Which C (profiles/r04_wide_tables.txt): what an MSM costs in the pipeline is sort + SMVP + the stitch / reduce work that runs beside the
// next launch, and that grows with the bucket sets -- 20 bits (16 of them) loses to the endomorphism mode at 2^20 although its SMVP is 0.85 ms
// alone against 0.99, and wins by 18 % at 2^24; 17 bits (2 of them) wins at 2^20.
// The digit width C is a template parameter of the two kernels (the tables are built for it when the bases are set: msm_hip.hip picks it from
// the number of bases -- 16 bits up to 2^16 points and 17 up to 2^20, where the bucket sets' stitch / reduce still counts, 20 beyond).
template <int C>
struct WideCfg {
  static_assert(C >= 16 && C <= 20, "digit bits of the wide tables");
  static constexpr int BITS = C;
  static constexpr int TABLES = WinCfg<C>::NWIN;  // 16 / 15 / 15 / 14 / 13 tables 2^(C w) P_i at 16 / 17 / 18 / 19 / 20 bits
  static constexpr int VWIN = 1 << (C - WBITS);   // 1 / 2 / 4 / 8 / 16 virtual windows of 2^15 slots
  static constexpr int KEYS = VWIN * NCOARSE;     // (virtual window, coarse bin) runs
  static_assert(VWIN <= MAXLW, "virtual windows are local windows");
};
__host__ __device__ constexpr int wide_tables_of(int bits) { return (254 + bits) / bits; }
__host__ __device__ constexpr int wide_vwin_of(int bits) { return 1 << (bits - WBITS); }
// The top digit.  The last window holds what is left of the scalar above bit C (T - 1) -- 16 / 7 / 14 bits at C = 17 / 19 / 20 -- so its
// magnitudes would all fall into the lowest virtual windows, which would then carry far more entries than the others, and the SMVP's lanes are
// as long as the fullest window makes them (first measurement at 20 bits: SMVP 1.06 ms instead of 0.85).  The top table is therefore
// 2^(C (T - 1) - top_shift) P_i and the top digit is used as d << top_shift: the same product for any point (exact integer arithmetic: no
// assumption on the point's order), spread over the virtual windows.  top_shift (msm_hip.hip: wide_top_shift) is the largest for which the top
// digit of every scalar below the scalar field's modulus stays within 2^(C-1): for BN254 0 / 11 / 5 at 17 / 19 / 20 bits.  A scalar whose
// shifted top digit passes that -- at or above the modulus -- is rejected like one that overflows the reference's recode (ERRBIT_SCALAR_CARRY).
// signed C-bit digit of window w of the biased scalar t (WinCfg<C>::WORDS words): its magnitude 1 .. 2^(C - 1) (0: no entry) and sign
template <int C>
__device__ __forceinline__ uint32_t wide_digit(const uint32_t* t, int w, int top_shift, uint32_t& sign, uint32_t& overflow) {
  constexpr uint32_t H = 1u << (C - 1);
  const int bit = C * w, i = bit >> 5, sh = bit & 31;
  uint32_t b = t[i] >> sh;
  if (sh + C > 32 && i + 1 < WinCfg<C>::WORDS) b |= t[i + 1] << (32 - sh);
  if (w == WideCfg<C>::TABLES - 1) {
    // the top digit: never negative (nothing above it carries into it), so its field is read with everything above it -- a digit of exactly
    // 2^(C-1), which the C-bit field cannot hold (17-bit digits of a scalar of 2^254 or more: Pallas, Vesta), is the bucket magnitude 2^(C-1) like
    // any other; beyond that, or beyond it after the shift, the scalar is rejected
    sign = 0;
    const uint32_t d = b - H;  // (b >= H: the bias bit of this window is set and the digit is not negative)
    if (d > (H >> top_shift)) {
      overflow = 1;
      return 0;
    }
    return d << top_shift;
  }
  b &= (1u << C) - 1u;
  sign = b < H ? 1u : 0u;
  return b >= H ? b - H : H - b;
}
// Magnitude m (1 .. 2^(C-1)) -> virtual window and bucket slot, INTERLEAVED (round 5):  vw = (m - 1) mod VWIN,  value(slot) = (m - 1) / VWIN + 1
// (1 .. 2^15; slot = value mod 2^15, i.e. slot 0 carries 2^15 as in every window).  Consecutive magnitudes go to consecutive virtual windows, so
// ANY smooth distribution of magnitudes -- the narrow top digit's included -- fills the virtual windows evenly: the shares of a window-sharded
// run (one virtual window per rank at 19 bits and 8 GPUs) are balanced, a whole MSM's windows need the same chunk length, and the top digit
// needs no shift.  (Rounds 4's contiguous ranges, vw = (m - 1) >> 15, put the whole top digit into the lowest windows; its shift spread it as
// multiples of 2^shift -- every 32nd slot of a 20-bit set four times as full as its neighbours, which a stitch wave pays for in all 64 lanes.)
// The window's weighted sum W_vw = sum_slot value(slot) B[slot] and plain total TC_vw give  sum_m m B_m = VWIN W_vw - (VWIN - 1 - vw) TC_vw.
template <int C>
__device__ __forceinline__ uint32_t wide_slot(uint32_t mag) { return (((mag - 1u) >> (C - WBITS)) + 1u) & 0x7fffu; }
template <int C>
__device__ __forceinline__ uint32_t wide_key(uint32_t mag) {  // (virtual window, coarse bin); mag = 0 gives garbage: callers test mag first
  return (((mag - 1u) & (uint32_t)(WideCfg<C>::VWIN - 1)) << 7) | (wide_slot<C>(mag) >> 8);
}

// first pass: counts[lw][tile][bin] (the layout of k_count), local window lw = v * VWIN + hi for scalar vector v of the launch's nvec
// (vec_stride words apart: several whole MSMs over the same tables share one kernel sequence, as in k_count)
// Virtual-window SHARES (round 5: the wide tables behind the window-sharded / multi-GPU entry points): a launch may take only the virtual
// windows [v_begin, v_begin + v_count) of every vector -- a rank of an 8-GPU run at 19-bit digits takes ONE of the 8: a bucket set of 2^15
// slots and, for uniform scalars, 14 n / 8 entries instead of the 2 n entries and two bucket sets of two 16-bit windows.  Both passes still
// recode every digit of every scalar (the carry chain runs across the digits; 32 B per scalar) and drop the digits whose magnitude falls
// outside the range; local window lw = v * v_count + (hi - v_begin).  Whole MSMs: v_begin = 0, v_count = VWIN.
template <int C>
__global__ void __launch_bounds__(256) k_count_wide(const uint32_t* __restrict__ scalars, size_t n, uint32_t tile_len, uint32_t tiles, int nvec,
                                                    size_t vec_stride, uint32_t* __restrict__ counts, uint32_t* __restrict__ err, int top_shift,
                                                    int v_begin, int v_count) {
  constexpr int SW = 8;  // full-length scalars
  constexpr int WIDE_KEYS = WideCfg<C>::KEYS, WIDE_TABLES = WideCfg<C>::TABLES;
  __shared__ uint32_t cnt[WIDE_KEYS];
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * tile_len;
  const size_t end = base + tile_len < n ? base + tile_len : n;
  uint32_t bad = 0;
  const int v = blockIdx.y;  // grid (tiles, nvec): one tile of one scalar vector per workgroup (as k_count)
  (void)nvec;
  const uint32_t keys = (uint32_t)v_count * NCOARSE;  // (virtual window, coarse bin) runs of this launch's share (<= WIDE_KEYS)
  for (int i = tid; i < WIDE_KEYS; i += 256) cnt[i] = 0;
  __syncthreads();
  for (size_t i0 = base; i0 < end; i0 += 256) {
    const size_t i = i0 + tid;
    if (i >= end) continue;
    uint32_t s[SW], tb[WinCfg<C, SW>::WORDS], t16[8], neg = 0;
    ld_scalar<SW>(scalars + (size_t)v * vec_stride + i * SW, s, neg);
    (void)bias_scalar<C, SW>(s, tb);  // (a top digit beyond the recode's range is caught where it is read: wide_digit)
    bad |= bias_scalar<16>(s, t16);   // the input contract of every mode: what overflows the reference's 16-bit recode is rejected (test/utils.rs:150-152)
#pragma unroll
    for (int w = 0; w < WIDE_TABLES; w++) {
      uint32_t sign;
      const uint32_t mag = wide_digit<C>(tb, w, top_shift, sign, bad);
      const uint32_t key = wide_key<C>(mag) - ((uint32_t)v_begin << 7);  // (mag = 0: no entry, whatever the key says)
      if (mag && key < keys) atomicAdd(&cnt[key], 1u);
    }
  }
  __syncthreads();
  for (int i = tid; i < (int)keys; i += 256)
    counts[((size_t)(v * v_count + i / NCOARSE) * tiles + blockIdx.x) * NCOARSE + (i % NCOARSE)] = cnt[i];
  if (bad) atomicOr(err, ERRBIT_SCALAR_CARRY);
}

// ---- shares of a few virtual windows: the first pass leaves a COMPACT LIST of the share's entries (round 5) ---------------------------------
// A rank of a window-sharded run keeps an eighth of the digits (one of 8 virtual windows at 19 bits, two of 16 at 20).  Ranking and staging
// them where they are found -- 13 x 8 digit positions per thread, an eighth of the lanes active at each -- made the second pass the longest
// kernel of the sort (310 - 540 us per launch of 8 vectors against 153 for the digit-plane scatter of two 16-bit windows,
// profiles/r05_wide_shares.txt).  So the divergent work is done ONCE, here: every kept digit is appended (wave-aggregated: one LDS atomic per
// wave, digit position and virtual window) to the list of its (local window, sub-tile of LIST_SUB scalars), and the second pass
// (k_scatter_list) reads the lists with every lane busy.
//   entry  = position within the sub-tile (11 bits) | table w << 11 | sign << 15 | bucket slot << 16   (the virtual window is the list's)
//   list of (lw, sub-tile q): list[lw * stride + q * LIST_SUB * TABLES ...], list_len[lw * subtiles + q] entries -- the arrays of the final
//   slot order (val_idxs), free until the fine sort writes them, sized for a share that receives every digit (stride >= n TABLES).
constexpr int LIST_SUB = 2048;
constexpr int WIDE_SHARE_VWIN_MAX = 4;  // shares of more virtual windows than this run the whole-MSM shape of the two passes
template <int C>
__global__ void __launch_bounds__(256) k_count_wide_list(const uint32_t* __restrict__ scalars, size_t n, uint32_t tile_len, uint32_t tiles, int nvec,
                                                         size_t vec_stride, uint32_t* __restrict__ counts, uint32_t* __restrict__ err, int top_shift,
                                                         int v_begin, int v_count, uint32_t* __restrict__ list, uint32_t* __restrict__ list_len,
                                                         size_t stride, uint32_t subtiles) {
  constexpr int SW = 8;
  constexpr int WIDE_TABLES = WideCfg<C>::TABLES;
  constexpr int KEYS_MAX = WIDE_SHARE_VWIN_MAX * NCOARSE;
  __shared__ uint32_t cnt[KEYS_MAX];
  __shared__ uint32_t lcount[WIDE_SHARE_VWIN_MAX];
  const int tid = threadIdx.x, lane = tid & 63;
  const size_t base = (size_t)blockIdx.x * tile_len;  // (tile_len is a multiple of LIST_SUB or the only tile's: sub-tiles never straddle tiles)
  const size_t end = base + tile_len < n ? base + tile_len : n;
  uint32_t bad = 0;
  const int v = blockIdx.y;
  (void)nvec;
  const uint32_t keys = (uint32_t)v_count * NCOARSE;
  for (int i = tid; i < KEYS_MAX; i += 256) cnt[i] = 0;
  if (tid < WIDE_SHARE_VWIN_MAX) lcount[tid] = 0;
  __syncthreads();
  for (size_t sub = base; sub < end; sub += LIST_SUB) {
    const size_t sub_end = sub + LIST_SUB < end ? sub + LIST_SUB : end;
    const uint32_t q = (uint32_t)(sub / LIST_SUB);
    for (size_t i0 = sub; i0 < sub_end; i0 += 256) {
      const size_t i = i0 + tid;
      const bool valid = i < sub_end;
      uint32_t s[SW], tb[WinCfg<C, SW>::WORDS], t16[8], neg = 0;
#pragma unroll
      for (int k = 0; k < SW; k++) s[k] = 0;  // (a lane beyond the end recodes zero: no entries)
      if (valid) ld_scalar<SW>(scalars + (size_t)v * vec_stride + i * SW, s, neg);
      (void)bias_scalar<C, SW>(s, tb);
      bad |= bias_scalar<16>(s, t16);   // the input contract of every mode (test/utils.rs:150-152)
      // every kept digit's entry and its place among the wave's kept digits of the same local window (ballots only: no LDS round trip) ...
      uint32_t ent[WIDE_TABLES], place[WIDE_TABLES];  // place: local window << 28 | position within the wave's block of that window; 0xffffffff: not kept
      uint32_t wave_cnt[WIDE_SHARE_VWIN_MAX];         // wave-uniform running counts
#pragma unroll
      for (int vw = 0; vw < WIDE_SHARE_VWIN_MAX; vw++) wave_cnt[vw] = 0;
#pragma unroll
      for (int w = 0; w < WIDE_TABLES; w++) {
        uint32_t sign;
        const uint32_t mag = wide_digit<C>(tb, w, top_shift, sign, bad);
        const uint32_t key = wide_key<C>(mag) - ((uint32_t)v_begin << 7);
        const bool keep = mag && key < keys;
        if (keep) atomicAdd(&cnt[key], 1u);
        ent[w] = (uint32_t)(i - sub) | ((uint32_t)w << 11) | (sign << 15) | (wide_slot<C>(mag) << 16);
        place[w] = 0xffffffffu;
#pragma unroll
        for (int vw = 0; vw < WIDE_SHARE_VWIN_MAX; vw++) {
          if (vw >= v_count) break;  // wave-uniform
          const bool mine = keep && (key >> 7) == (uint32_t)vw;
          const unsigned long long mm = __ballot(mine);
          if (mine) place[w] = ((uint32_t)vw << 28) | (wave_cnt[vw] + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull)));
          wave_cnt[vw] += (uint32_t)__popcll(mm);
        }
      }
      // ... ONE reservation per wave and local window for all of them (one LDS round trip instead of one per digit position), then the stores
      uint32_t wave_base[WIDE_SHARE_VWIN_MAX];
#pragma unroll
      for (int vw = 0; vw < WIDE_SHARE_VWIN_MAX; vw++) {
        wave_base[vw] = 0;
        if (vw < v_count && lane == 0 && wave_cnt[vw]) wave_base[vw] = atomicAdd(&lcount[vw], wave_cnt[vw]);
      }
#pragma unroll
      for (int vw = 0; vw < WIDE_SHARE_VWIN_MAX; vw++) wave_base[vw] = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_base[vw]);
#pragma unroll
      for (int w = 0; w < WIDE_TABLES; w++) {
        if (place[w] != 0xffffffffu) {
          const uint32_t vw = place[w] >> 28;
          const uint32_t b = vw == 0 ? wave_base[0] : vw == 1 ? wave_base[1] : vw == 2 ? wave_base[2] : wave_base[3];
          list[(size_t)(v * v_count + vw) * stride + (size_t)q * (LIST_SUB * WIDE_TABLES) + b + (place[w] & 0x0fffffffu)] = ent[w];
        }
      }
    }
    __syncthreads();
    if (tid < v_count) {
      list_len[(size_t)(v * v_count + tid) * subtiles + q] = lcount[tid];
      lcount[tid] = 0;
    }
    __syncthreads();
  }
  for (int i = tid; i < (int)keys; i += 256)
    counts[((size_t)(v * v_count + i / NCOARSE) * tiles + blockIdx.x) * NCOARSE + (i % NCOARSE)] = cnt[i];
  if (bad) atomicOr(err, ERRBIT_SCALAR_CARRY);
}

// second pass of a share: grid (tiles, local windows) -- a workgroup takes the lists of ONE local window over its tile, LIST_CHUNK entries at a
// time: histogram of the coarse bins, scan, cursor placement into the LDS staging, coalesced write-out (k_scatter_coarse's scheme with every lane
// busy).
constexpr int LIST_CHUNK = 4096;
__global__ void __launch_bounds__(256) k_scatter_list(const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_len, size_t stride, uint32_t region,
                                                      uint32_t subtiles, size_t n, uint32_t tile_len, uint32_t tiles, int w_eff,
                                                      const uint32_t* __restrict__ counts, const uint32_t* __restrict__ bin_total,
                                                      uint32_t* __restrict__ coarse_ptr, uint32_t* __restrict__ tmp_val, uint8_t* __restrict__ tmp_fine,
                                                      size_t table_stride, uint32_t chunks, uint32_t host_chunk_len, uint32_t* __restrict__ chunk_len_dev) {
  __shared__ uint32_t gpos[NCOARSE];
  __shared__ uint32_t hist[NCOARSE];
  __shared__ uint32_t lstart[NCOARSE];
  __shared__ uint32_t cur[NCOARSE];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t st_val[LIST_CHUNK];
  __shared__ uint32_t st_dst[LIST_CHUNK];
  __shared__ uint8_t st_fine[LIST_CHUNK];
  __shared__ uint32_t max_total;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lw = blockIdx.y;
  if (tid == 0) max_total = 0;
  __syncthreads();
  // start of every coarse bin's run of this tile (exclusive scan of the window's 128 bin totals + what earlier tiles put there); workgroup
  // (0, 0) does it for every local window of the launch (its own last): it publishes all bin starts and the launch's chunk length
  const bool publisher = blockIdx.x == 0 && blockIdx.y == 0;
  for (int pl = publisher ? w_eff - 1 : lw; pl >= lw; pl--) {
    const int bin = tid;  // threads 0 .. 127: one bin each (two waves)
    const bool live = tid < NCOARSE;
    const uint32_t v = live ? bin_total[pl * NCOARSE + bin] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    const uint32_t incl = x + ((wid & 1) ? wave_tot[wid - 1] : 0u);
    if (live) gpos[bin] = incl - v + counts[((size_t)pl * tiles + blockIdx.x) * NCOARSE + bin];
    if (live && publisher) {
      coarse_ptr[(size_t)pl * (NCOARSE + 1) + bin] = incl - v;
      if (bin == NCOARSE - 1) {
        coarse_ptr[(size_t)pl * (NCOARSE + 1) + NCOARSE] = incl;
        atomicMax(&max_total, incl);
      }
    }
    __syncthreads();
  }
  if (publisher && tid == 0) *chunk_len_dev = smvp_chunk_len(max_total, chunks, host_chunk_len);
  const size_t tile_base = (size_t)blockIdx.x * tile_len;
  const size_t tile_end = tile_base + tile_len < n ? tile_base + tile_len : n;
  uint32_t* ov = tmp_val + (size_t)lw * stride;
  uint8_t* of = tmp_fine + (size_t)lw * stride;
  for (size_t sub = tile_base; sub < tile_end; sub += LIST_SUB) {
    const uint32_t q = (uint32_t)(sub / LIST_SUB);
    const uint32_t len = list_len[(size_t)lw * subtiles + q];
    const uint32_t* src = list + (size_t)lw * stride + (size_t)q * region;
    for (uint32_t lo = 0; lo < len; lo += LIST_CHUNK) {
      const uint32_t cnt = len - lo < (uint32_t)LIST_CHUNK ? len - lo : (uint32_t)LIST_CHUNK;
      if (tid < NCOARSE) hist[tid] = 0;
      __syncthreads();
      uint32_t ent[LIST_CHUNK / 256];
#pragma unroll
      for (int j = 0; j < LIST_CHUNK / 256; j++) {
        const uint32_t e = (uint32_t)j * 256 + tid;
        ent[j] = e < cnt ? src[lo + e] : 0xffffffffu;     // (a slot is 15 bits: no entry has bit 31 set)
        if (e < cnt) atomicAdd(&hist[ent[j] >> 24], 1u);  // coarse bin = slot >> 8 = entry >> 24
      }
      __syncthreads();
      const uint32_t mine = tid < NCOARSE ? hist[tid] : 0u;
      const uint32_t excl = block_excl_scan_256(mine, wave_tot);
      if (tid < NCOARSE) {
        lstart[tid] = excl;
        cur[tid] = excl;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < LIST_CHUNK / 256; j++) {
        if (ent[j] != 0xffffffffu) {
          const uint32_t bin = ent[j] >> 24;
          const uint32_t e = atomicAdd(&cur[bin], 1u);
          // window w of point i = record w * n_bases + i
          st_val[e] = ((uint32_t)(((ent[j] >> 11) & 15u) * table_stride) + (uint32_t)sub + (ent[j] & 2047u)) | (((ent[j] >> 15) & 1u) << 31);
          st_fine[e] = (uint8_t)((ent[j] >> 16) & 0xffu);
          st_dst[e] = gpos[bin] + (e - lstart[bin]);
        }
      }
      __syncthreads();
      for (uint32_t e = tid; e < cnt; e += 256) {
        const uint32_t d = st_dst[e];
        ov[d] = st_val[e];
        of[d] = st_fine[e];
      }
      if (tid < NCOARSE) gpos[tid] += hist[tid];
      __syncthreads();
    }
  }
}

// second pass: the LDS-ranked, LDS-staged scatter of k_scatter_coarse over all (virtual window, coarse bin) runs at once -- 256 / 1024 / 2048
// of them at 17 / 19 / 20 bits.  ALL digits of the 2048 scalars of a block iteration are staged together (30 720 / 28 672 / 26 624 entries:
// 120 / 28 / 13 per run): ranked per window as k_scatter_coarse does, a run would receive a fraction of that per iteration and every 4-byte
// store would be a memory transaction of its own.  One workgroup of 512 threads per CU.  (With 1024 scalars per iteration the kernel took
// 244 / 351 / 471 us at 2^22 points: the shorter the runs, the worse the stores coalesce.)
//
// Two shapes of the same kernel (WideShape<C, SHARE>):
//   whole MSMs   512 threads, 4 scalars per thread and iteration (3 at 16 bits), every run of the bucket set, LDS for every entry the iteration's
//                scalars can produce (153 - 158 KB: one workgroup per CU)
//   shares       (round 5: a rank's virtual windows, k_count_wide) -- at most 4 virtual windows, an eighth of the entries for uniform scalars at 8
//                ranks: with the whole-MSM shape the 4096 workgroups of a launch of 8 vectors ran one per CU, sixteen rounds of a latency-bound
//                kernel (385 - 544 us per launch against 153 for the digit-plane scatter of two 16-bit windows, profiles/r05_wide_shares.txt).
//                256 threads, 8 scalars per thread, LDS for WIDE_SHARE_CAP entries (38 KB: four workgroups per CU), no ranks in registers.  Skewed
//                scalars may put EVERY digit of an iteration into the share (14 x 2048 entries): an iteration whose entries pass the staging is
//                redone one scalar per thread at a time.
#ifndef WIDE_SHARE_REREAD
#define WIDE_SHARE_REREAD 1  // A/B aid (same-box pairs, profiles/r05_wide_shares.txt: 0.2198 - 0.2231 vs 0.2242 - 0.2256 ms per MSM share): 1 = the scalars are read again for the second pass (one at a time) instead of staying in registers
#endif
constexpr int WIDE_THREADS = 512;
constexpr int WIDE_SHARE_THREADS = 256, WIDE_SHARE_CAP = 6144;
template <int C, bool SHARE>
struct WideScatterShape {
  static constexpr int THREADS = SHARE ? WIDE_SHARE_THREADS : WIDE_THREADS;
  static constexpr int PER = SHARE ? 8 : (C == 16 ? 3 : 4);                    // scalars per thread and block iteration
  static constexpr int SUB = THREADS * PER;                                    // scalars staged per block iteration
  static constexpr int KEYS = SHARE ? (WideCfg<C>::VWIN < WIDE_SHARE_VWIN_MAX ? WideCfg<C>::VWIN : WIDE_SHARE_VWIN_MAX) * NCOARSE : WideCfg<C>::KEYS;
  static constexpr int STAGE = SHARE ? WIDE_SHARE_CAP : SUB * WideCfg<C>::TABLES;  // entries the LDS staging holds
  static_assert(SUB <= 2048 && WideCfg<C>::TABLES <= 16, "sign | window | position in 16 bits");
  static_assert(STAGE * 5 + KEYS * (SHARE ? 16 : 12) + 64 <= 160 * 1024, "LDS of a workgroup");
};
template <int C, bool SHARE>
__global__ void __launch_bounds__((WideScatterShape<C, SHARE>::THREADS), (SHARE ? (WIDE_SHARE_REREAD ? 4 : 3) : 1)) k_scatter_wide(const uint32_t* __restrict__ scalars, size_t n, size_t stride, uint32_t tile_len,
                                                               uint32_t tiles, int nvec, size_t vec_stride, const uint32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ bin_total, uint32_t* __restrict__ coarse_ptr,
                                                               uint32_t* __restrict__ tmp_val, uint8_t* __restrict__ tmp_fine, size_t table_stride,
                                                               uint32_t chunks, uint32_t host_chunk_len, uint32_t* __restrict__ chunk_len_dev, int top_shift,
                                                               int v_begin, int v_count) {
  using Shape = WideScatterShape<C, SHARE>;
  constexpr int SW = 8;  // full-length scalars
  constexpr int WIDE_KEYS = Shape::KEYS, WIDE_TABLES = WideCfg<C>::TABLES, THREADS = Shape::THREADS;
  const int keys = v_count * NCOARSE;                // runs of this launch's share of the virtual windows (k_count_wide); WIDE_KEYS for whole MSMs
  const uint32_t key0 = (uint32_t)v_begin << 7;
  // 5 bytes of LDS per staged entry -- its (virtual window, coarse bin) run, its fine slot, and sign | window | position within the iteration's
  // scalars (16 bits: the record index is put together when the entry is written out) -- so that 2048 scalars (1536 at 16 bits) fit one
  // iteration: twice the run length of the 4-byte index staged before (153 - 158 KB of the 160 KB a workgroup may hold)
  constexpr int WIDE_PER = Shape::PER;
  constexpr int WIDE_SUB = Shape::SUB;      // scalars staged per block iteration
  constexpr int WIDE_STAGE = Shape::STAGE;  // entries staged at a time
  __shared__ uint32_t gpos[WIDE_KEYS];    // write cursor of every run of this tile, relative to its virtual window's array
  __shared__ uint32_t hist[WIDE_KEYS];
  __shared__ uint32_t lstart[WIDE_KEYS];
  __shared__ uint32_t cur[SHARE ? WIDE_KEYS : 1];  // share shape: cursor of every run while an iteration's entries are staged
  __shared__ uint16_t st_loc[WIDE_STAGE];
  __shared__ uint16_t st_key[WIDE_STAGE];
  __shared__ uint8_t st_fine[WIDE_STAGE];
  __shared__ uint32_t wave_tot[THREADS / 64];
  __shared__ uint32_t max_total;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) max_total = 0;
  __syncthreads();
  const size_t tile_base = (size_t)blockIdx.x * tile_len;
  const size_t tile_end = tile_base + tile_len < n ? tile_base + tile_len : n;
  // grid (tiles, nvec): a workgroup scatters one tile of ONE scalar vector (one MSM of the launch), whose v_count local windows start at lw0.
  // Start of every run: exclusive scan of each virtual window's 128 bin totals (a pair of waves per window) + what
  // earlier tiles put there.  Workgroup (0, 0) does this for every vector of the launch (its own last: gpos keeps the last one scanned): it
  // publishes all bin starts (coarse_ptr[lw][0 .. 128]) and the launch's chunk length.
  const bool publisher = blockIdx.x == 0 && blockIdx.y == 0;
  const int lw0 = (int)blockIdx.y * v_count;
  const uint32_t* sv = scalars + (size_t)blockIdx.y * vec_stride;
  for (int pv = publisher ? nvec - 1 : (int)blockIdx.y; pv >= (int)blockIdx.y; pv--)
  for (int i0 = 0; i0 < keys; i0 += THREADS) {
    const int i = i0 + tid, lw = pv * v_count + i / NCOARSE, bin = i % NCOARSE;
    const bool live = i < keys;  // (fewer runs than threads: 17-bit digits, shares of a few virtual windows)
    const uint32_t v = live ? bin_total[lw * NCOARSE + bin] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    const uint32_t incl = x + ((wid & 1) ? wave_tot[wid - 1] : 0u);
    if (live) gpos[i] = incl - v + counts[((size_t)lw * tiles + blockIdx.x) * NCOARSE + bin];
    if (live && publisher) {
      coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin] = incl - v;
      if (bin == NCOARSE - 1) {
        coarse_ptr[(size_t)lw * (NCOARSE + 1) + NCOARSE] = incl;
        atomicMax(&max_total, incl);
      }
    }
    __syncthreads();
  }
  if (publisher && tid == 0) *chunk_len_dev = smvp_chunk_len(max_total, chunks, host_chunk_len);
  // this thread's scalar j of the block iteration at `sub`, biased for the recode.  The scalars are read twice -- for the counts and for the
  // entries (the second time from the L2) --: held in registers across the scan they and the ranks passed the 256 registers a wave may have
  auto biased = [&](size_t sub, int j, uint32_t* tb) {
    const size_t i = sub + (size_t)j * THREADS + tid;
    uint32_t raw[SW], neg = 0;
#pragma unroll
    for (int k = 0; k < SW; k++) raw[k] = 0;  // an all-zero scalar recodes to all-zero digits: no entries
    if (i < tile_end) ld_scalar<SW>(sv + i * SW, raw, neg);
    (void)bias_scalar<C, SW>(raw, tb);
  };
  // exclusive scan of the run lengths hist[] -> lstart[]: KPT consecutive keys per thread (with fewer runs than threads, the first WIDE_KEYS threads
  // take one each); ends with a barrier
  auto scan_runs = [&]() {
    constexpr int KPT = WIDE_KEYS >= THREADS ? WIDE_KEYS / THREADS : 1;
    static_assert(KPT * THREADS == WIDE_KEYS || WIDE_KEYS < THREADS, "keys per thread");
    const bool mine = KPT * tid < WIDE_KEYS;
    uint32_t h[KPT], sum = 0;
#pragma unroll
    for (int k = 0; k < KPT; k++) {
      h[k] = mine ? hist[KPT * tid + k] : 0u;
      sum += h[k];
    }
    uint32_t x = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    uint32_t run = x - sum;
    for (int k = 0; k < wid; k++) run += wave_tot[k];
#pragma unroll
    for (int k = 0; k < KPT; k++) {
      if (mine) lstart[KPT * tid + k] = run;
      run += h[k];
    }
    __syncthreads();
  };
  // the staged entries 0 .. cnt-1 (bin-major) to their runs; ends with a barrier
  auto write_out = [&](size_t sub, uint32_t cnt) {
    for (uint32_t e = tid; e < cnt; e += THREADS) {
      const uint32_t key = st_key[e];
      const size_t d = (size_t)(lw0 + (key >> 7)) * stride + gpos[key] + (e - lstart[key]);
      const uint32_t loc = st_loc[e];
      // window w of point i = record w * n_bases + i
      tmp_val[d] = ((uint32_t)(((loc >> 11) & 15u) * table_stride) + (uint32_t)sub + (loc & 2047u)) | ((loc >> 15) << 31);
      tmp_fine[d] = st_fine[e];
    }
    __syncthreads();
  };
  for (size_t sub = tile_base; sub < tile_end; sub += WIDE_SUB) {
    if constexpr (SHARE) {
      // Share shape: no ranks are kept -- an entry's place inside its run is drawn from a cursor when it is staged (any order inside a run is as
      // good as another) -- and the iteration's biased scalars stay in registers across both passes (8 x 9 words; their 8 loads are in flight
      // together): nothing is read twice, and the LDS alone bounds the workgroups per CU.
#if WIDE_SHARE_REREAD
#define WIDE_SHARE_UNROLL _Pragma("unroll 1")
#define WIDE_SHARE_TB(j) tb1
#define WIDE_SHARE_LOAD(j) uint32_t tb1[WinCfg<C, SW>::WORDS]; biased(sub, j, tb1)
#else
#define WIDE_SHARE_UNROLL _Pragma("unroll")
#define WIDE_SHARE_TB(j) tbs[j]
#define WIDE_SHARE_LOAD(j)
      uint32_t tbs[WIDE_PER][WinCfg<C, SW>::WORDS];
#pragma unroll
      for (int j = 0; j < WIDE_PER; j++) biased(sub, j, tbs[j]);
#endif
      for (int k = tid; k < WIDE_KEYS; k += THREADS) hist[k] = 0;
      __syncthreads();
      WIDE_SHARE_UNROLL
      for (int j = 0; j < WIDE_PER; j++) {
        WIDE_SHARE_LOAD(j);
#pragma unroll
        for (int w = 0; w < WIDE_TABLES; w++) {
          uint32_t sign, over = 0;
          const uint32_t mag = wide_digit<C>(WIDE_SHARE_TB(j), w, top_shift, sign, over);  // (an overflowing top digit: no entry here as in k_count_wide, which reports it)
          const uint32_t key = wide_key<C>(mag) - key0;                            // (outside this launch's virtual windows: no entry)
          if (mag && key < (uint32_t)keys) atomicAdd(&hist[key], 1u);
        }
        __builtin_amdgcn_sched_barrier(0);  // one scalar's digits at a time: hoisted together, the 8 x 14 digits and keys take 400 registers
      }
      __syncthreads();
      scan_runs();
      const uint32_t total = lstart[WIDE_KEYS - 1] + hist[WIDE_KEYS - 1];
      if (total <= (uint32_t)WIDE_STAGE) {  // block-uniform
        for (int k = tid; k < WIDE_KEYS; k += THREADS) cur[k] = lstart[k];
        __syncthreads();
        // (the digits are extracted AGAIN from the biased scalars: kept from the counting pass -- which is what the compiler does when it can
        //  see that the values are the same -- the 8 x 14 magnitudes, keys and signs take 400 registers; the asm makes the words opaque)
#if !WIDE_SHARE_REREAD
#pragma unroll
        for (int j = 0; j < WIDE_PER; j++)
#pragma unroll
          for (int k = 0; k < WinCfg<C, SW>::WORDS; k++) asm volatile("" : "+v"(tbs[j][k]));
#endif
        WIDE_SHARE_UNROLL
        for (int j = 0; j < WIDE_PER; j++) {
          WIDE_SHARE_LOAD(j);
#pragma unroll
          for (int w = 0; w < WIDE_TABLES; w++) {
            uint32_t sign, over = 0;
            const uint32_t mag = wide_digit<C>(WIDE_SHARE_TB(j), w, top_shift, sign, over);
            const uint32_t key = wide_key<C>(mag) - key0;
            if (mag && key < (uint32_t)keys) {
              const uint32_t e = atomicAdd(&cur[key], 1u);
              st_loc[e] = (uint16_t)((sign << 15) | ((uint32_t)w << 11) | (uint32_t)(j * THREADS + tid));
              st_key[e] = (uint16_t)key;
              st_fine[e] = (uint8_t)(wide_slot<C>(mag) & 0xffu);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        write_out(sub, total);
        for (int k = tid; k < WIDE_KEYS; k += THREADS) gpos[k] += hist[k];
        __syncthreads();
      } else {
        // Skewed scalars put more of the iteration's digits into this share than the staging holds (every digit, at worst): the iteration is
        // redone one scalar per thread at a time (THREADS x TABLES entries at most), each read again (the rare path keeps nothing in registers)
        static_assert(THREADS * WIDE_TABLES <= WIDE_STAGE, "one scalar per thread fits the staging");
        __syncthreads();  // (everyone has read `total` before hist / lstart are rebuilt)
#pragma unroll 1
        for (int j = 0; j < WIDE_PER; j++) {
          for (int k = tid; k < WIDE_KEYS; k += THREADS) hist[k] = 0;
          __syncthreads();
          uint32_t tb[WinCfg<C, SW>::WORDS];
          biased(sub, j, tb);
#pragma unroll 1
          for (int w = 0; w < WIDE_TABLES; w++) {
            uint32_t sign, over = 0;
            const uint32_t mag = wide_digit<C>(tb, w, top_shift, sign, over);
            const uint32_t key = wide_key<C>(mag) - key0;
            if (mag && key < (uint32_t)keys) atomicAdd(&hist[key], 1u);
          }
          __syncthreads();
          scan_runs();
          const uint32_t part = lstart[WIDE_KEYS - 1] + hist[WIDE_KEYS - 1];
          for (int k = tid; k < WIDE_KEYS; k += THREADS) cur[k] = lstart[k];
          __syncthreads();
#pragma unroll 1
          for (int w = 0; w < WIDE_TABLES; w++) {
            uint32_t sign, over = 0;
            const uint32_t mag = wide_digit<C>(tb, w, top_shift, sign, over);
            const uint32_t key = wide_key<C>(mag) - key0;
            if (mag && key < (uint32_t)keys) {
              const uint32_t e = atomicAdd(&cur[key], 1u);
              st_loc[e] = (uint16_t)((sign << 15) | ((uint32_t)w << 11) | (uint32_t)(j * THREADS + tid));
              st_key[e] = (uint16_t)key;
              st_fine[e] = (uint8_t)(wide_slot<C>(mag) & 0xffu);
            }
          }
          __syncthreads();
          write_out(sub, part);
          for (int k = tid; k < WIDE_KEYS; k += THREADS) gpos[k] += hist[k];
          __syncthreads();
        }
      }
    } else {
    for (int k = tid; k < WIDE_KEYS; k += THREADS) hist[k] = 0;
    __syncthreads();
    uint32_t rank[WIDE_PER][(WIDE_TABLES + 1) / 2];  // two 16-bit ranks per register (a run holds fewer than 2^16 entries)
#pragma unroll
    for (int j = 0; j < WIDE_PER; j++) {
      uint32_t tb[WinCfg<C, SW>::WORDS];
      biased(sub, j, tb);
#pragma unroll
      for (int w = 0; w < WIDE_TABLES; w++) {
        uint32_t sign, over = 0;
        const uint32_t mag = wide_digit<C>(tb, w, top_shift, sign, over);  // (an overflowing top digit: no entry here as in k_count_wide, which reports it)
        const uint32_t key = wide_key<C>(mag) - key0;                           // (outside this launch's virtual windows: no entry)
        const uint32_t r = mag && key < (uint32_t)keys ? atomicAdd(&hist[key], 1u) : 0u;
        if (w & 1) rank[j][w >> 1] |= r << 16;
        else rank[j][w >> 1] = r;
      }
    }
    __syncthreads();
    scan_runs();
    const uint32_t total = lstart[WIDE_KEYS - 1] + hist[WIDE_KEYS - 1];
#pragma unroll
    for (int j = 0; j < WIDE_PER; j++) {
      uint32_t tb[WinCfg<C, SW>::WORDS];
      biased(sub, j, tb);
#pragma unroll
      for (int w = 0; w < WIDE_TABLES; w++) {
        uint32_t sign, over = 0;
        const uint32_t mag = wide_digit<C>(tb, w, top_shift, sign, over);
        const uint32_t key = wide_key<C>(mag) - key0;
        if (mag && key < (uint32_t)keys) {
          const uint32_t e = lstart[key] + ((rank[j][w >> 1] >> ((w & 1) * 16)) & 0xffffu);
          st_loc[e] = (uint16_t)((sign << 15) | ((uint32_t)w << 11) | (uint32_t)(j * THREADS + tid));
          st_key[e] = (uint16_t)key;
          st_fine[e] = (uint8_t)(wide_slot<C>(mag) & 0xffu);
        }
      }
    }
    __syncthreads();
    write_out(sub, total);
    for (int k = tid; k < WIDE_KEYS; k += THREADS) gpos[k] += hist[k];
    __syncthreads();
    }
  }
}

// The second pass of a launch whose first pass left digit planes (k_count with negbits != null): the same LDS-ranked, LDS-staged
// scatter as k_scatter_coarse, reading 2 B per (input, local window) from the planes.  No scalar arithmetic and no scalars in
// registers.  `w_eff` local windows of `w_count_vec` windows per scalar vector.  negbits == null: input `pos` is scalar `pos` and multiplies
// base `pos`.  Endomorphism halves (k_count<C, 4, true>): input 2 j + h is half h of scalar j, its sign bit j of negbits[v][h], and it
// multiplies record j + h * half_shift (half_shift = n_bases).
__global__ void __launch_bounds__(256) k_scatter_planes(const uint16_t* __restrict__ planes, const uint64_t* __restrict__ negbits, size_t n,
                                                        size_t stride, uint32_t tile_len, uint32_t tiles, int w_eff, int w_count_vec,
                                                        const uint32_t* __restrict__ counts, const uint32_t* __restrict__ bin_total,
                                                        uint32_t* __restrict__ coarse_ptr, uint32_t* __restrict__ tmp_val,
                                                        uint8_t* __restrict__ tmp_fine, uint32_t half_shift,
                                                        uint32_t chunks, uint32_t host_chunk_len, uint32_t* __restrict__ chunk_len_dev) {
  __shared__ uint32_t gpos[MAXLW * NCOARSE];
  __shared__ uint32_t hist[NCOARSE];
  __shared__ uint32_t lstart[NCOARSE];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t st_val[SCAT_SUB];
  __shared__ uint32_t st_dst[SCAT_SUB];
  __shared__ uint8_t st_fine[SCAT_SUB];
  __shared__ uint32_t max_total;
  const int tid = threadIdx.x;
  if (tid == 0) max_total = 0;
  __syncthreads();
  for (int i0 = 0; i0 < w_eff * NCOARSE; i0 += 256) {  // bin starts of every local window: as k_scatter_coarse
    const int i = i0 + tid, lw = i / NCOARSE, bin = i % NCOARSE, lane = tid & 63;
    const bool live = i < w_eff * NCOARSE;
    const uint32_t v = live ? bin_total[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[tid >> 6] = x;
    __syncthreads();
    const uint32_t incl = x + ((tid >> 6) & 1 ? wave_tot[(tid >> 6) - 1] : 0u);
    if (live) gpos[i] = incl - v + counts[((size_t)lw * tiles + blockIdx.x) * NCOARSE + bin];
    if (live && blockIdx.x == 0) {
      coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin] = incl - v;
      if (bin == NCOARSE - 1) {
        coarse_ptr[(size_t)lw * (NCOARSE + 1) + NCOARSE] = incl;
        atomicMax(&max_total, incl);
      }
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) *chunk_len_dev = smvp_chunk_len(max_total, chunks, host_chunk_len);
  const size_t tile_base = (size_t)blockIdx.x * tile_len;
  const size_t tile_end = tile_base + tile_len < n ? tile_base + tile_len : n;
  const size_t neg_words = (n / 2 + 63) / 64;
  for (int lw = 0; lw < w_eff; lw++) {
    const uint16_t* pl = planes + (size_t)lw * n;
    const uint64_t* nb = negbits ? negbits + (size_t)(lw / w_count_vec) * 2 * neg_words : nullptr;
    uint32_t* ov = tmp_val + (size_t)lw * stride;
    uint8_t* of = tmp_fine + (size_t)lw * stride;
    for (size_t sub = tile_base; sub < tile_end; sub += SCAT_SUB) {
      uint32_t code[8];
      uint32_t negs = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const size_t i = sub + (size_t)j * 256 + tid;
        code[j] = i < tile_end ? pl[i] : 0u;
        if (nb && i < tile_end) negs |= (uint32_t)((nb[(i & 1) * neg_words + (i >> 1) / 64] >> ((i >> 1) & 63)) & 1ull) << j;
      }
      if (tid < NCOARSE) hist[tid] = 0;
      __syncthreads();
      uint32_t rank[8];
#pragma unroll
      for (int j = 0; j < 8; j++) rank[j] = code[j] ? atomicAdd(&hist[(code[j] & 0x7fffu) >> 8], 1u) : 0u;
      __syncthreads();
      const uint32_t mine = tid < NCOARSE ? hist[tid] : 0u;
      const uint32_t excl = block_excl_scan_256(mine, wave_tot);
      if (tid < NCOARSE) lstart[tid] = excl;
      __syncthreads();
      const uint32_t total = lstart[NCOARSE - 1] + hist[NCOARSE - 1];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (code[j]) {
          const uint32_t slot = code[j] & 0x7fffu, bin = slot >> 8;
          const uint32_t e = lstart[bin] + rank[j];
          uint32_t pos = (uint32_t)(sub + (size_t)j * 256 + tid);
          if (nb) pos = (pos >> 1) + ((pos & 1u) ? half_shift : 0u);
          st_val[e] = pos | (((code[j] >> 15) ^ ((negs >> j) & 1u)) << 31);
          st_fine[e] = (uint8_t)(slot & 0xffu);
          st_dst[e] = gpos[lw * NCOARSE + bin] + rank[j];
        }
      }
      __syncthreads();
      for (uint32_t e = tid; e < total; e += 256) {
        const uint32_t d = st_dst[e];
        ov[d] = st_val[e];
        of[d] = st_fine[e];
      }
      if (tid < NCOARSE) gpos[lw * NCOARSE + tid] += hist[tid];
      __syncthreads();
    }
  }
}

#ifndef MSM_FINE_CHUNK
#define MSM_FINE_CHUNK 4096
#endif
constexpr int FINE_CHUNK = MSM_FINE_CHUNK;  // entries staged per block iteration (16 per thread; 8192 -- runs of 32 entries per slot -- measured slower:
                                            // 1.55 -> 1.69 ms at 2^24, 0.092 -> 0.106 at 2^20, profiles/r05_sort.txt)
constexpr int FINE_PER = FINE_CHUNK / 256;  // entries per thread and iteration

// counter[key] += 1 for every active lane, returning the lane's rank (old value).  All lanes that share the key of the
// wave's first active lane are served by ONE LDS atomic (ballot + popcount): with heavily skewed scalars (many equal
// digits) nearly the whole wave shares a key and a plain ds_add would serialise 64-fold; with uniform digits this costs
// one ballot.  Must be called with the same `valid` pattern by whole waves (inactive lanes pass valid = false).
__device__ __forceinline__ uint32_t lds_count_rank(uint32_t* counter, uint32_t key, bool valid) {
  const unsigned long long vm = __ballot(valid);
  if (vm == 0) return 0;
  const int first = __ffsll((long long)vm) - 1;  // wave-uniform: v_readlane, no LDS round trip
  const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
  const bool same = valid && key == k0;
  const unsigned long long sm = __ballot(same);
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == first) base = atomicAdd(&counter[k0], (uint32_t)__popcll(sm));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
  uint32_t rank = base + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
  if (valid && !same) rank = atomicAdd(&counter[key], 1u);
  return rank;
}
// the same without the rank: no atomic has to return, so consecutive calls do not wait for each other
__device__ __forceinline__ void lds_count_only(uint32_t* counter, uint32_t key, bool valid) {
  const unsigned long long vm = __ballot(valid);
  if (vm == 0) return;
  const int first = __ffsll((long long)vm) - 1;
  const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
  const bool same = valid && key == k0;
  const unsigned long long sm = __ballot(same);
  if ((int)(threadIdx.x & 63) == first) atomicAdd(&counter[k0], (uint32_t)__popcll(sm));
  if (valid && !same) atomicAdd(&counter[key], 1u);
}

// A coarse bin with more than FINE_BIG entries (heavily skewed scalars: e.g. all entries of a window in one slot) is shared
// by FINE_SPLIT workgroups WITHOUT any cross-block communication: each of them histograms the whole bin (1 byte per entry)
// and, in the same sweep, the part in front of its own contiguous sub-range -- that gives it the start of every slot and
// its own offset inside every slot -- and then scatters only its sub-range.  Normal bins are handled by workgroup 0 alone
// (the other FINE_SPLIT - 1 exit at once).
constexpr int FINE_SPLIT = 8;

constexpr uint32_t FINE_BIG = 32768;  // (a multiple of FINE_CHUNK)
static_assert(FINE_BIG % FINE_CHUNK == 0, "sub-ranges are whole chunks");

// Histograms of the FINE_SPLIT sub-ranges of every coarse bin that exceeds FINE_BIG (part_hist[lw][bin][part][256]); launched
// ahead of k_sort_fine when n is large enough for uniform scalars to produce such bins (the host decides), so that the
// FINE_SPLIT workgroups of a bin do not each histogram the whole bin.  Smaller bins: nothing to do.
__global__ void __launch_bounds__(256) k_fine_hist(const uint8_t* __restrict__ tmp_fine, size_t stride,
                                                   const uint32_t* __restrict__ coarse_ptr, uint32_t* __restrict__ part_hist) {
  __shared__ uint32_t hist[FINE];
  const int bin = blockIdx.x, part = blockIdx.z, lw = blockIdx.y, tid = threadIdx.x;
  const uint32_t begin = coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin], end = coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin + 1];
  if (end - begin <= FINE_BIG) return;
  uint32_t per = (end - begin + FINE_SPLIT - 1) / FINE_SPLIT;
  per = (per + FINE_CHUNK - 1) / FINE_CHUNK * FINE_CHUNK;
  const uint32_t my_begin = begin + (uint32_t)part * per < end ? begin + (uint32_t)part * per : end;
  const uint32_t my_end = my_begin + per < end ? my_begin + per : end;
  const uint8_t* tf = tmp_fine + (size_t)lw * stride;
  hist[tid] = 0;
  __syncthreads();
  for (uint32_t base = my_begin; base < my_end; base += FINE_CHUNK) {
    uint32_t f[FINE_PER];
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      const uint32_t i = base + j * 256 + tid;
      f[j] = i < my_end ? tf[i] : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) lds_count_only(hist, f[j] & 0xffu, f[j] != 0xffffffffu);
  }
  __syncthreads();
  part_hist[(((size_t)lw * NCOARSE + bin) * FINE_SPLIT + part) * FINE + tid] = hist[tid];
}

__global__ void __launch_bounds__(256) k_sort_fine(const uint32_t* __restrict__ tmp_val, const uint8_t* __restrict__ tmp_fine, size_t stride,
                                                   const uint32_t* __restrict__ coarse_ptr, uint32_t* __restrict__ col_ptr,
                                                   uint32_t* __restrict__ val_idxs, uint32_t chunks, const uint32_t* __restrict__ chunk_len_dev,
                                                   uint32_t* __restrict__ chunk_slot, const uint32_t* __restrict__ part_hist, uint32_t* __restrict__ info) {
  const uint32_t chunk_len = *chunk_len_dev;
  __shared__ uint32_t hist[FINE];
  __shared__ uint32_t before[FINE];  // entries of every slot in front of this workgroup's sub-range
  __shared__ uint32_t lstart[FINE];
  __shared__ uint32_t gpos[FINE];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t st_val[FINE_CHUNK];
  __shared__ uint8_t st_slot[FINE_CHUNK];  // (round 5: an entry's destination is its run's cursor + its place in the staged run -- recomputed at the
                                           //  write-out from the slot, 1 B, instead of staged as 4 B: 24.6 KB instead of 36.9 -- six workgroups per CU, not four)
  __shared__ uint32_t long_c0[FINE], long_c1[FINE], long_slot[FINE];  // (the chunk table's long runs: at most one per slot)
  __shared__ uint32_t skew_flag, long_count;
  const int bin = blockIdx.x, part = blockIdx.z, lw = blockIdx.y, tid = threadIdx.x;
  const uint32_t half = gridDim.x * FINE;  // bucket slots per window: the grid covers exactly the window's coarse bins
  const uint32_t begin = coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin], end = coarse_ptr[(size_t)lw * (NCOARSE + 1) + bin + 1];
  const bool big = end - begin > FINE_BIG;
  if (!big && part != 0) return;
  // this workgroup's sub-range [my_begin, my_end): the whole bin, or one of FINE_SPLIT pieces (multiples of FINE_CHUNK)
  uint32_t my_begin = begin, my_end = end;
  if (big) {
    uint32_t per = (end - begin + FINE_SPLIT - 1) / FINE_SPLIT;
    per = (per + FINE_CHUNK - 1) / FINE_CHUNK * FINE_CHUNK;
    my_begin = begin + (uint32_t)part * per < end ? begin + (uint32_t)part * per : end;
    my_end = my_begin + per < end ? my_begin + per : end;
  }
  const uint32_t* tv = tmp_val + (size_t)lw * stride;
  const uint8_t* tf = tmp_fine + (size_t)lw * stride;
  uint32_t* out = val_idxs + (size_t)lw * stride;
  // pass 1: slot histogram of the whole coarse bin (and of the part in front of the sub-range)
  hist[tid] = 0;
  before[tid] = 0;
  if (tid == 0) {
    skew_flag = 0;
    long_count = 0;
  }
  __syncthreads();
  if (!big) {
    for (uint32_t base = begin; base < end; base += FINE_CHUNK) {  // 16 independent byte loads in flight per thread
      uint32_t f[FINE_PER];
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) {
        const uint32_t i = base + j * 256 + tid;
        f[j] = i < end ? tf[i] : 0xffffffffu;
      }
#pragma unroll
      for (int j = 0; j < FINE_PER; j++)
        if (f[j] != 0xffffffffu) atomicAdd(&hist[f[j]], 1u);
    }
  } else if (part_hist) {
    // the sub-range histograms were made by k_fine_hist: sum them (and the ones in front of this workgroup's sub-range)
    if (tid == 0 && part == 0) atomicOr(info, INFOBIT_HUGE_BIN);
    const uint32_t* ph = part_hist + ((size_t)lw * NCOARSE + bin) * FINE_SPLIT * FINE + tid;
    uint32_t all = 0, front = 0;
#pragma unroll
    for (int q = 0; q < FINE_SPLIT; q++) {
      const uint32_t c = ph[q * FINE];
      all += c;
      if (q < part) front += c;
    }
    hist[tid] = all;
    before[tid] = front;
  } else {
    if (tid == 0 && part == 0) atomicOr(info, INFOBIT_HUGE_BIN);  // (a huge bin without k_fine_hist's histograms: every sharer sweeps the bin up to its own end)
    // FINE_CHUNK entries per sweep step, 16 independent byte loads per thread in flight; a step lies wholly in front of
    // the sub-range or not (my_begin - begin is a multiple of FINE_CHUNK), so every entry is counted once
    uint32_t f[FINE_PER], g[FINE_PER];  // double buffered: the loads of step k + 1 are in flight while step k is counted
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      const uint32_t i = begin + j * 256 + tid;
      f[j] = i < end ? tf[i] : 0xffffffffu;
    }
    for (uint32_t base = begin; base < end; base += FINE_CHUNK) {
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) {
        const uint32_t i = base + FINE_CHUNK + j * 256 + tid;
        g[j] = i < end ? tf[i] : 0xffffffffu;
      }
      uint32_t* counter = base < my_begin ? before : hist;
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) lds_count_only(counter, f[j] & 0xffu, f[j] != 0xffffffffu);
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) f[j] = g[j];
    }
    __syncthreads();
    hist[tid] += before[tid];
  }
  __syncthreads();
  // a slot holding more than a quarter of the bin means skewed scalars: pass 2 then ranks with wave-aggregated atomics
  if (hist[tid] > (end - begin) / 4 && end - begin > (uint32_t)FINE_CHUNK) skew_flag = 1;
  {
    const uint32_t excl = block_excl_scan_256(hist[tid], wave_tot);
    gpos[tid] = begin + excl + before[tid];
    if (part == 0) {
      col_ptr[(size_t)lw * (half + 1) + bin * FINE + tid] = begin + excl;
      if (bin == (int)gridDim.x - 1 && tid == FINE - 1) col_ptr[(size_t)lw * (half + 1) + half] = end;
    }
    // SMVP chunks whose first entry lies in this slot's run [first, last): short runs are tabulated by their own thread
    // (of workgroup 0), long ones (skewed scalars) by all threads of all workgroups of the bin together
    const uint32_t first = begin + excl, last = first + hist[tid];
    uint32_t c0 = (first + chunk_len - 1) / chunk_len;
    uint32_t c1 = (uint32_t)(((uint64_t)last + chunk_len - 1) / chunk_len);
    if (c1 > chunks) c1 = chunks;
    if (c0 > c1) c0 = c1;
    const bool long_run = c1 - c0 > 16;
    if (part == 0 && !long_run)
      for (uint32_t c = c0; c < c1; c++) chunk_slot[(size_t)lw * chunks + c] = (uint32_t)(bin * FINE + tid);
    if (long_run) {
      const uint32_t k = atomicAdd(&long_count, 1u);
      long_c0[k] = c0;
      long_c1[k] = c1;
      long_slot[k] = (uint32_t)(bin * FINE + tid);
    }
  }
  __syncthreads();
  {
    const uint32_t nl = long_count, nparts = big ? FINE_SPLIT : 1;
    for (uint32_t k = 0; k < nl; k++)
      for (uint32_t c = long_c0[k] + part * 256 + tid; c < long_c1[k]; c += nparts * 256) chunk_slot[(size_t)lw * chunks + c] = long_slot[k];
  }
  __syncthreads();
  // pass 2: LDS-staged scatter of the sub-range, FINE_CHUNK entries at a time
  const bool skewed = skew_flag != 0;  // block-uniform (read after the barriers of the scan above)
  for (uint32_t base = my_begin; base < my_end; base += FINE_CHUNK) {
    hist[tid] = 0;
    __syncthreads();
    uint32_t v[FINE_PER], fr[FINE_PER];  // value; slot | rank << 8
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      const uint32_t i = base + j * 256 + tid;
      const bool valid = i < my_end;
      const uint32_t f = valid ? tf[i] : 0u;
      const uint32_t rank = skewed ? lds_count_rank(hist, f, valid) : (valid ? atomicAdd(&hist[f], 1u) : 0u);
      if (valid) {
        v[j] = tv[i];
        fr[j] = f | (rank << 8);
      } else {
        fr[j] = 0xffffffffu;
      }
    }
    __syncthreads();
    const uint32_t excl = block_excl_scan_256(hist[tid], wave_tot);
    lstart[tid] = excl;
    __syncthreads();
    const uint32_t total = (my_end - base) < (uint32_t)FINE_CHUNK ? (my_end - base) : (uint32_t)FINE_CHUNK;
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      if (fr[j] != 0xffffffffu) {
        const uint32_t f = fr[j] & 0xffu, r = fr[j] >> 8;
        const uint32_t e = lstart[f] + r;
        st_val[e] = v[j];
        st_slot[e] = (uint8_t)f;
      }
    }
    __syncthreads();
    for (uint32_t e = tid; e < total; e += 256) {
      const uint32_t f = st_slot[e];
      out[gpos[f] + (e - lstart[f])] = st_val[e];
    }
    __syncthreads();
    gpos[tid] += hist[tid];
    __syncthreads();
  }
}

// Deterministic mode of the transpose (the reference's stage test asserts the exact val_idxs,
// tests/transpose_shader.rs:198-199): the order inside a slot is the arrival order of LDS atomics -- the group sum does not depend on it,
// but a stage-level comparison does.  With the debug switch on (msm_hip_set_debug), every slot's run is put into ascending order of its
// entries (index | sign << 31: the positive digits' points by index, then the negative digits') by a rank sort: one lane per entry finds
// its slot (binary search of col_ptr), counts the entries of its run that are smaller (they are distinct) and writes itself to that
// position of a scratch copy (`tmp`, the coarse-order array, free by then); a second kernel copies the scratch back.  O(sum of run
// length^2): runs beyond ORDER_RUN_MAX entries (heavily skewed inputs) keep their arrival order.
constexpr uint32_t ORDER_RUN_MAX = 1u << 15;  // (2^30 comparisons for one such run)
__global__ void __launch_bounds__(256) k_order_runs(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ val_idxs, uint32_t* __restrict__ tmp,
                                                    size_t stride, uint32_t half) {
  const int lw = blockIdx.y;
  const uint32_t* cp = col_ptr + (size_t)lw * (half + 1);
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= cp[half]) return;
  uint32_t lo = 0, hi = half - 1;  // the slot whose run holds entry e: cp[s] <= e < cp[s + 1]
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cp[mid + 1] > e) hi = mid;
    else lo = mid + 1;
  }
  const uint32_t b = cp[lo], end = cp[lo + 1];
  const uint32_t* v = val_idxs + (size_t)lw * stride;
  const uint32_t x = v[e];
  uint32_t pos = e;
  if (end - b <= ORDER_RUN_MAX) {
    uint32_t rank = 0;
    for (uint32_t j = b; j < end; j++) rank += v[j] < x ? 1u : 0u;
    pos = b + rank;
  }
  tmp[(size_t)lw * stride + pos] = x;
}
__global__ void __launch_bounds__(256) k_copy_runs(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ tmp, uint32_t* __restrict__ val_idxs,
                                                   size_t stride, uint32_t half) {
  const int lw = blockIdx.y;
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e < col_ptr[(size_t)lw * (half + 1) + half]) val_idxs[(size_t)lw * stride + e] = tmp[(size_t)lw * stride + e];
}

// ------------------------------------------------------------------------------------------------ stage 3: SMVP
// Bucket accumulate (≙ smvp.template.wgsl:31-117, CPU model test/utils.rs:166-219):
//   B[w][k] = sum_{d=+k} P - sum_{d=-k} P  (k >= 1),   B[w][0] = -sum_{d=-2^15} P
// The reference gives one thread one bucket, so a wave runs as long as its fullest bucket.  Here every lane owns a
// fixed-length chunk of `chunk_len` consecutive entries of the slot-sorted list -- equal work per lane whatever the bucket
// sizes -- and flushes its accumulator whenever the slot changes.  The host picks a chunk length (in
// [SMVP_CHUNK_MIN, SMVP_CHUNK_MAX]; msm_hip.hip: chunk_len_for) so that about SMVP_TARGET_LANES lanes exist for n entries per window (about three rounds of 3 waves
// per SIMD at 168 VGPRs, no scratch) and sizes the chunk arrays and grids with it; the length actually used is settled on the device
// from the entries the sort produced (smvp_chunk_len above: never longer than the host's).  Runs that cross a chunk boundary leave a "tail" piece
// (in the chunk where the run starts) and "head" pieces (in the chunks it continues into); k_smvp_stitch adds them.
// Buckets and pieces are stored as raw XYZZ records (no multiplication on the flush path).
constexpr int SMVP_CHUNK_MIN = SMVP_CHUNK_MIN_ENTRIES;
constexpr int SMVP_CHUNK_MAX = 1024;
constexpr int SMVP_TARGET_LANES = 9 << 16;  // three rounds of 3 waves per SIMD (1024 SIMDs x 64 lanes).  Round 3 sweep (profiles/r03_lanes_sweep.txt): against two
                                             // rounds the kernel itself is 3 % faster (shorter chunks even out the SIMDs' finishing times), the stitch has 1.5 x the pieces to
                                             // add, and the step is equal or up to 2 % shorter (2^18, plain bases, window shares); four rounds and more lose to the stitch
constexpr int REC_WORDS = (XYZZ_WORDS + 3) / 4 * 4;  // 160 B record with 9 limbs: 36 limbs, valid flag, 3 pad words; 16-byte aligned (240 B with 14)
constexpr int REC_FLAG = 4 * FQ_L;                   // word index of the valid flag

// word I of a record: limbs of x, y, zz, zzz, then the valid flag, then padding (static indices only: an intermediate word array would
// not be promoted to registers and cost the stitch / row-column kernels a 148-byte scratch object each)
template <int I>
__device__ __forceinline__ uint32_t rec_get(const g1_xyzz& a) {
  if constexpr (I < FQ_L) return a.x.v[I];
  else if constexpr (I < 2 * FQ_L) return a.y.v[I - FQ_L];
  else if constexpr (I < 3 * FQ_L) return a.zz.v[I - 2 * FQ_L];
  else if constexpr (I < 4 * FQ_L) return a.zzz.v[I - 3 * FQ_L];
  else if constexpr (I == REC_FLAG) return a.inf ? 0u : 1u;
  else return 0u;
}
template <int I>
__device__ __forceinline__ void rec_set(g1_xyzz& a, uint32_t v) {
  if constexpr (I < FQ_L) a.x.v[I] = v;
  else if constexpr (I < 2 * FQ_L) a.y.v[I - FQ_L] = v;
  else if constexpr (I < 3 * FQ_L) a.zz.v[I - 2 * FQ_L] = v;
  else if constexpr (I < 4 * FQ_L) a.zzz.v[I - 3 * FQ_L] = v;
}
template <int... K>
__device__ __forceinline__ void st_rec_quads(uint4* q, const g1_xyzz& a, std::integer_sequence<int, K...>) {
  ((q[K] = make_uint4(rec_get<4 * K>(a), rec_get<4 * K + 1>(a), rec_get<4 * K + 2>(a), rec_get<4 * K + 3>(a))), ...);
}
template <int... K>
__device__ __forceinline__ void ld_rec_quads(const uint4* q, g1_xyzz& a, std::integer_sequence<int, K...>) {
  ((void)([&] {
     const uint4 v = q[K];
     rec_set<4 * K>(a, v.x);
     rec_set<4 * K + 1>(a, v.y);
     rec_set<4 * K + 2>(a, v.z);
     rec_set<4 * K + 3>(a, v.w);
   }()),
   ...);
}
__device__ __forceinline__ void st_rec(uint32_t* p, const g1_xyzz& a) {
  st_rec_quads(reinterpret_cast<uint4*>(p), a, std::make_integer_sequence<int, REC_WORDS / 4>{});
}
__device__ __forceinline__ g1_xyzz ld_rec(const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 fl = q[REC_FLAG / 4];
  const uint32_t flag = REC_FLAG % 4 == 0 ? fl.x : (REC_FLAG % 4 == 1 ? fl.y : (REC_FLAG % 4 == 2 ? fl.z : fl.w));
  if (flag == 0) return g1_identity();
  g1_xyzz a;
  ld_rec_quads(q, a, std::make_integer_sequence<int, REC_WORDS / 4>{});
  a.inf = false;
  return a;
}

// Assignments to the SMVP loop's accumulator from its rare paths (a run's first point; the repair after a doubling).  On the device they
// are written as moves INTO the accumulator's own registers (read-write operands): a plain assignment makes every coordinate a merge of
// two values at the end of the branch, and the register allocator then keeps the merged value in the rare path's registers -- the hot
// path, which updates the coordinates in place, pays a copy per limb and iteration to get there and back (measured: 36 + 18 v_mov).
__device__ __forceinline__ void smvp_set(fq& dst, const fq& src) {
#pragma unroll
  for (int i = 0; i < FQ_L; i++) asm("v_mov_b32 %0, %1" : "+v"(dst.v[i]) : "v"(src.v[i]));
}
__device__ __forceinline__ void smvp_set_one(fq& dst) {
#pragma unroll
  for (int i = 0; i < FQ_L; i++) asm("v_mov_b32 %0, %1" : "+v"(dst.v[i]) : "s"(FQ_ONE29[i]));
}
__device__ __forceinline__ void smvp_restart(g1_xyzz& acc, const fq& px, const fq& py) {
  smvp_set(acc.x, px);
  smvp_set(acc.y, py);
  smvp_set_one(acc.zz);
  smvp_set_one(acc.zzz);
  acc.inf = false;
}
__device__ __forceinline__ void smvp_assign(g1_xyzz& acc, const g1_xyzz& src) {
  smvp_set(acc.x, src.x);
  smvp_set(acc.y, src.y);
  smvp_set(acc.zz, src.zz);
  smvp_set(acc.zzz, src.zzz);
  acc.inf = src.inf;
}

// 168 VGPRs hold the 9-limb loop (3 waves per SIMD); 14 limbs take up to 256 (2 waves).  Fq2 on 9 limbs (BN254 G2) wants 278: capped at 256 -- 12 words
// of scratch -- because the second wave is worth 23 % of the kernel (4.29 -> 3.31 ms at 2^20, profiles/r03_g2_throughput.txt); Fq2 on 14 limbs: one wave
#ifndef MSM_SMVP_WAVES_WIDE
#define MSM_SMVP_WAVES_WIDE 1  // Fq2 on 14 limbs (BLS12-381 G2): 256 VGPRs + 130 AGPRs at one wave; two waves = a 256-register cap with scratch (A/B: profiles/r04_g2_two_waves.txt)
#endif
#ifndef MSM_SMVP_WAVES_9
#define MSM_SMVP_WAVES_9 3  // (2: experiment -- a third of the register file and of the wave slots left to the kernels of another launch, profiles/r05_two_context_overlap.txt)
#endif
constexpr int SMVP_WAVES_PER_SIMD = FQ_L <= 9 ? MSM_SMVP_WAVES_9 : FQ_L <= 18 ? 2 : MSM_SMVP_WAVES_WIDE;
// the stitch and the row / column sums (full additions: the widest kernels after the SMVP) in a unit with 18 limbs per coordinate: two waves per
// SIMD as well (256 VGPRs + 0.26 KB of scratch instead of 309 - 317 + AGPRs: -1.5 % per MSM)
#ifndef MSM_REDUCE_WAVES_FQ2
#define MSM_REDUCE_WAVES_FQ2 2
#endif
constexpr int REDUCE_WAVES_PER_SIMD = FQ_L == 18 ? MSM_REDUCE_WAVES_FQ2 : 1;  // (1: no constraint beyond the workgroup size)
__global__ void __launch_bounds__(256, SMVP_WAVES_PER_SIMD) k_smvp_chunks(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ col_ptr,
                                                     const uint32_t* __restrict__ val_idxs, size_t stride, uint32_t chunks,
                                                     const uint32_t* __restrict__ chunk_len_dev, const uint32_t* __restrict__ chunk_slot,
                                                     uint32_t* __restrict__ buckets, uint32_t* __restrict__ heads,
                                                     uint32_t* __restrict__ tails, uint32_t half) {
  const uint32_t chunk_len = *chunk_len_dev;
#if defined(__HIP_DEVICE_COMPILE__) && MSM_SMVP_WAVES_9 == 2
  if constexpr (FQ_L <= 9) asm volatile("; two waves per SIMD" ::: "v175");  // 176 VGPRs allocated: 2 waves per SIMD whatever the loop needs
#endif
  const int lw = blockIdx.y;
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  const uint32_t* cp = col_ptr + (size_t)lw * (half + 1);
  const uint32_t nw = cp[half];
  const uint64_t begin64 = (uint64_t)c * chunk_len;
  if (c >= chunks || begin64 >= nw) return;
  const uint32_t begin = (uint32_t)begin64;
  const uint32_t end = (nw - begin > chunk_len) ? begin + chunk_len : nw;
  // slot containing entry `begin` (cp[s] <= begin < cp[s + 1]), tabulated by k_sort_fine
  uint32_t s = chunk_slot[(size_t)lw * chunks + c], run_begin = cp[s], run_end = cp[s + 1];
  const uint32_t* vi = val_idxs + (size_t)lw * stride;
  const size_t rec = ((size_t)lw * chunks + c) * REC_WORDS;
  g1_xyzz acc = g1_identity();
  bool wneg = false;  // sign carried by acc.y (g1_madd_w); applied when the accumulator is flushed
  // Two entries ahead: the index of entry t + 2 and the point of entry t + 1 are requested before the addition of entry t starts, so that
  // a gather that misses the Infinity Cache (bases beyond 256 MiB) is covered by ~4 us of arithmetic instead of stalling the wave.  Both
  // loads are unconditional (clamped to the chunk's last entry): a conditional load makes every prefetch register a merge of old and
  // new value, i.e. a copy per register and iteration.
  const uint32_t last = end - 1;
  uint32_t vnext = vi[begin];
  uint32_t vnn = vi[begin + 1 < end ? begin + 1 : last];
  uint32_t wx[CW], wy[CW];
  ld_coord(bases + (size_t)(vnext & 0x7fffffffu) * PT_WORDS, wx);
  ld_coord(bases + (size_t)(vnext & 0x7fffffffu) * PT_WORDS + CW, wy);
  for (uint32_t t = begin; t < end; t++) {
    const uint32_t v = vnext;
    fq px = fq_unpack(wx), py = fq_unpack(wy);
#if defined(__HIP_DEVICE_COMPILE__)
    // the loads below stay behind the unpacking (the limbs are pinned in front of this point): they can then reuse the registers of
    // wx / wy; hoisted above it they need a second set and 16 copies per iteration
#pragma unroll
    for (int i = 0; i < FQ_L; i++) asm volatile("" : "+v"(px.v[i]), "+v"(py.v[i]) : : "memory");
#endif
    vnext = vnn;
    vnn = vi[t + 2 < end ? t + 2 : last];
    {
      const uint32_t* pt = bases + (size_t)(vnext & 0x7fffffffu) * PT_WORDS;
      ld_coord(pt, wx);
      ld_coord(pt + CW, wy);
    }
    const bool sneg = (v >> 31) != 0u;  // bit 31: the digit is negative
    if (t == run_end) {  // the run of slot s ended inside this chunk
      if (run_begin >= begin) st_rec(buckets + ((size_t)lw * half + s) * REC_WORDS, g1_unsigned(acc, wneg));
      else st_rec(heads + rec, g1_unsigned(acc, wneg));
      acc.inf = true;  // (the coordinates of an empty accumulator are never read: no need to zero 36 registers)
      run_begin = run_end;
      // next non-empty slot (exists: t < nw).  Usually the very next one; after a few empty slots switch to a binary search of the
      // slot whose run contains entry t: a lane that walks thousands of empty slots with dependent loads (a few heavy buckets far
      // apart: few distinct scalars, or the two halves of equal scalars) would hold up the whole kernel for milliseconds
      int gap = 0;
      do { s++; run_end = cp[s + 1]; } while (run_end == run_begin && ++gap < 8);
      if (run_end == run_begin) {
        uint32_t lo = s + 1, hi = half - 1;  // cp[hi + 1] = nw > t = run_begin
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (cp[mid + 1] > run_begin) hi = mid;
          else lo = mid + 1;
        }
        s = lo;
        run_end = cp[s + 1];
      }
    }
    if (acc.inf) {  // the first point of a run: W = y with the digit's sign as the state -- no negation, no arithmetic
      smvp_restart(acc, px, py);
      wneg = sneg;
    } else {
      const int status = g1_madd_w_hot(acc, wneg, px, py, sneg);
      if (status) {  // the point met itself or its negative in the accumulator (duplicate bases): repair, from the point read again
        wneg = false;
        if (status == 1) {
          const uint32_t* pt = bases + (size_t)(v & 0x7fffffffu) * PT_WORDS;
          const fq qy = ld_fq(pt + CW);
          smvp_assign(acc, g1_double_affine(ld_fq(pt), sneg ? fq_neg_canonical(qy) : qy));
        } else {
          acc.inf = true;
        }
      }
    }
  }
  acc = g1_unsigned(acc, wneg);
  // last run of the chunk: complete only if it started here and ends exactly at or before `end`
  if (run_begin >= begin && run_end <= end) {
    st_rec(buckets + ((size_t)lw * half + s) * REC_WORDS, acc);
  } else if (run_begin < begin) {
    st_rec(heads + rec, acc);  // continuation of a run from an earlier chunk (it may continue further)
  } else {
    st_rec(tails + rec, acc);  // run starts here and continues into the next chunk(s)
  }
}

// x[dst] += x[src] for XYZZ records held in LDS
__device__ __forceinline__ void lds_add_pair(uint32_t* x, int dst, int src) {
  st_xyzz(x + dst * XYZZ_WORDS, g1_add(ld_xyzz(x + dst * XYZZ_WORDS), ld_xyzz(x + src * XYZZ_WORDS)));
}

// One lane per bucket slot: empty slots get the identity record (no memset of the bucket array is needed), runs that lie
// inside one chunk were already written by k_smvp_chunks, and a run that spans chunks c0 < ... < c1 is the tail piece of
// c0 plus the head pieces of c0+1 .. c1.  Buckets with more than STITCH_BIG pieces (heavily skewed scalars: one bucket
// may hold every entry of a window) are queued for k_smvp_stitch_big instead of being walked by one lane.
#ifndef MSM_STITCH_SORTED
#define MSM_STITCH_SORTED 1
#endif
constexpr uint32_t STITCH_BIG = 32;
constexpr uint32_t STITCH_BIG_CAP = 1 << 15;  // queue capacity; more big buckets than this fall back to the serial walk

__global__ void __launch_bounds__(256, REDUCE_WAVES_PER_SIMD) k_smvp_stitch(const uint32_t* __restrict__ col_ptr, uint32_t chunks, const uint32_t* __restrict__ chunk_len_dev,
                                                     const uint32_t* __restrict__ heads, const uint32_t* __restrict__ tails,
                                                     uint32_t* __restrict__ buckets, uint32_t* __restrict__ big_queue) {
  const int lw = blockIdx.y;
  const uint32_t half = gridDim.x * 256;              // bucket slots per window: one lane per slot
  const uint32_t chunk_len = *chunk_len_dev;
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;  // < half by grid construction
  const uint32_t* cp = col_ptr + (size_t)lw * (half + 1);
  const uint32_t b = cp[s], e = cp[s + 1];
  uint32_t c0 = 0, adds = 0;  // adds = c1 - c0 of a run this kernel walks, 0 for every other slot
  if (b == e) {
    st_rec(buckets + ((size_t)lw * half + s) * REC_WORDS, g1_identity());
  } else {
    c0 = b / chunk_len;
    const uint32_t c1 = (e - 1) / chunk_len;
    if (c0 != c1) {
      adds = c1 - c0;
      if (adds >= STITCH_BIG) {
        const uint32_t at = atomicAdd(&big_queue[0], 1u);
        if (at < STITCH_BIG_CAP) {
          big_queue[1 + at] = ((uint32_t)lw << 16) | s;
          adds = 0;
        }
      }
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // The queue pointer above is a scalar load issued inside a branch; in the units whose group addition is large (Fq2) the branches below are far
  // ones, expanded through a scavenged SGPR pair -- which must not have that load still in flight (tools/check_long_branch_hazard.py)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if MSM_STITCH_SORTED
  // Round 5: a slot's run spans 1 .. 5 chunks (64 entries per bucket against ~28 per chunk), and a wave walks as long as its longest run -- about 4.5
  // additions for a mean of 2.3.  The workgroup's 256 slots are therefore handed to its lanes in DESCENDING order of their addition count (a
  // counting sort over 8 classes through LDS): every wave then holds runs of nearly equal length, and the lanes with nothing to add fill the
  // last wave(s), which leave at once.
  __shared__ uint32_t cls[4][8];
  __shared__ uint32_t perm_s[256], perm_c0[256], perm_adds[256];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const uint32_t k = adds < 7u ? adds : 7u;
  unsigned long long mine = 0;
#pragma unroll
  for (uint32_t v = 0; v < 8; v++) {
    const unsigned long long bal = __ballot(k == v);
    if (k == v) mine = bal;
    if (lane == 0) cls[wave][v] = (uint32_t)__popcll(bal);
  }
  __syncthreads();
  uint32_t pos = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));  // same class, same wave, lower lane
#pragma unroll
  for (int w = 0; w < 4; w++) {
#pragma unroll
    for (uint32_t v = 0; v < 8; v++) {
      const uint32_t cnt = cls[w][v];
      if (v > k || (v == k && w < wave)) pos += cnt;  // longer runs first; same length: earlier waves first
    }
  }
  perm_s[pos] = s;
  perm_c0[pos] = c0;
  perm_adds[pos] = adds;
  __syncthreads();
  const uint32_t ms = perm_s[t], mc0 = perm_c0[t], madds = perm_adds[t];
#else
  const uint32_t ms = s, mc0 = c0, madds = adds;
#endif
  if (madds == 0) return;
  g1_xyzz acc = ld_rec(tails + ((size_t)lw * chunks + mc0) * REC_WORDS);
  for (uint32_t c = mc0 + 1; c <= mc0 + madds; c++) acc = g1_add(acc, ld_rec(heads + ((size_t)lw * chunks + c) * REC_WORDS));
  st_rec(buckets + ((size_t)lw * half + ms) * REC_WORDS, acc);
}

// Queued big buckets: one block per bucket (blocks stride over the queue); every thread adds a strided subset of the
// pieces, then an LDS tree.  The queue counter is reset for the slot's next launch by k_bpr_final, a later kernel of the same
// stream (a last-block hand-off here would cost a __threadfence per launch, queue empty or not).
// HUGE buckets (>= STITCH_HUGE pieces: one value shared by a large part of the scalars -- the ones and small constants of witness
// vectors, all-equal scalars) would keep their single workgroup busy for pieces / 256 dependent additions per thread while the rest
// of the GPU idles: when the queue is short (<= 256 items) the huge ones are shared by STITCH_BLOCKS / (their number) workgroups
// each; a workgroup leaves its partial sum in the slot's scratch records and the last one to arrive adds the partials
// (device-scope fences: only on this rare path).
// big_queue layout (words): [0] count, [1 .. CAP] items, [BIGQ_CHUNK_LEN] the launch's SMVP chunk length (smvp_chunk_len),
// [BIGQ_COUNTERS ..] 256 arrival counters (zero between launches), [BIGQ_SCRATCH ..] 256 XYZZ records.
constexpr uint32_t STITCH_HUGE = 1024;
constexpr int STITCH_BLOCKS = 256;  // grid of k_smvp_stitch_big
constexpr size_t BIGQ_CHUNK_LEN = STITCH_BIG_CAP + 1;
constexpr size_t BIGQ_COUNTERS = STITCH_BIG_CAP + 4;
constexpr size_t BIGQ_SCRATCH = BIGQ_COUNTERS + STITCH_BLOCKS;
constexpr size_t BIGQ_WORDS = BIGQ_SCRATCH + (size_t)STITCH_BLOCKS * REC_WORDS;

__global__ void __launch_bounds__(256) k_smvp_stitch_big(const uint32_t* __restrict__ col_ptr, uint32_t chunks,
                                                         const uint32_t* __restrict__ heads, const uint32_t* __restrict__ tails,
                                                         uint32_t* __restrict__ buckets, uint32_t* big_queue,
                                                         uint32_t half) {
  __shared__ uint32_t x[256 * XYZZ_WORDS];
  __shared__ uint32_t huge_flag[256], huge_list[256], wave_tot[4], s_nhuge, s_last;
  const int t = threadIdx.x;
  uint32_t count = big_queue[0];
  if (count > STITCH_BIG_CAP) count = STITCH_BIG_CAP;
  if (count == 0) return;
  const uint32_t chunk_len = big_queue[BIGQ_CHUNK_LEN];
  // the pieces of queue item `item`: chunks c0 .. c1 of local window lw, bucket slot s
  auto decode = [&](uint32_t item, uint32_t& lw, uint32_t& s, uint32_t& c0, uint32_t& c1) {
    const uint32_t code = big_queue[1 + item];
    lw = code >> 16;
    s = code & 0xffffu;
    const uint32_t* cp = col_ptr + (size_t)lw * (half + 1);
    c0 = cp[s] / chunk_len;
    c1 = (cp[s + 1] - 1) / chunk_len;
  };
  // block-wide sum of the per-thread accumulators -> x[0]
  auto block_sum = [&](const g1_xyzz& acc) {
    st_xyzz(x + t * XYZZ_WORDS, acc);
    __syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
      if (t < sft) lds_add_pair(x, t, t + sft);
      __syncthreads();
    }
  };
  // which items are huge (every workgroup computes the same list, in queue order)
  const bool may_split = count <= 256 && gridDim.x == (unsigned)STITCH_BLOCKS;
  uint32_t nhuge = 0;
  if (may_split) {
    uint32_t f = 0;
    if ((uint32_t)t < count) {
      uint32_t lw, s, c0, c1;
      decode(t, lw, s, c0, c1);
      f = c1 - c0 + 1 >= STITCH_HUGE ? 1u : 0u;
    }
    huge_flag[t] = f;
    const uint32_t pos = block_excl_scan_256(f, wave_tot);
    if (f) huge_list[pos] = (uint32_t)t;
    if (t == 255) s_nhuge = pos + f;
    __syncthreads();
    nhuge = s_nhuge;
  }
  for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    if (may_split && huge_flag[item]) continue;  // block-uniform
    uint32_t lw, s, c0, c1;
    decode(item, lw, s, c0, c1);
    g1_xyzz acc = g1_identity();
    for (uint32_t c = c0 + t; c <= c1; c += 256) {
      const uint32_t* piece = (c == c0 ? tails : heads) + ((size_t)lw * chunks + c) * REC_WORDS;
      acc = g1_add(acc, ld_rec(piece));
    }
    block_sum(acc);
    if (t == 0) st_rec(buckets + ((size_t)lw * half + s) * REC_WORDS, ld_xyzz(x));
    __syncthreads();
  }
  if (nhuge == 0) return;
  const uint32_t per = (uint32_t)STITCH_BLOCKS / nhuge;  // workgroups per huge bucket, >= 1
  const uint32_t hi = blockIdx.x / per, sub = blockIdx.x % per;
  if (hi >= nhuge) return;
  uint32_t lw, s, c0, c1;
  decode(huge_list[hi], lw, s, c0, c1);
  g1_xyzz acc = g1_identity();
  for (uint32_t c = c0 + sub * 256 + t; c <= c1; c += per * 256) {
    const uint32_t* piece = (c == c0 ? tails : heads) + ((size_t)lw * chunks + c) * REC_WORDS;
    acc = g1_add(acc, ld_rec(piece));
  }
  block_sum(acc);
  uint32_t* out = buckets + ((size_t)lw * half + s) * REC_WORDS;
  if (per == 1) {
    if (t == 0) st_rec(out, ld_xyzz(x));
    return;
  }
  uint32_t* counters = big_queue + BIGQ_COUNTERS;
  uint32_t* scratch = big_queue + BIGQ_SCRATCH;
  if (t == 0) {
    st_rec(scratch + (size_t)blockIdx.x * REC_WORDS, ld_xyzz(x));
    __threadfence();  // the partial is visible device-wide before this workgroup is counted
    s_last = atomicAdd(&counters[hi], 1u) == per - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire: the other workgroups' partials
  block_sum((uint32_t)t < per ? ld_rec(scratch + ((size_t)hi * per + t) * REC_WORDS) : g1_identity());
  if (t == 0) {
    st_rec(out, ld_xyzz(x));
    counters[hi] = 0;  // ready for the slot's next launch
  }
}

// ------------------------------------------------------------------------------------------------ stage 4: bucket reduce
// S_w = sum_{k=1}^{h-1} k * B[k] + h * B[0]   (≙ bpr.template.wgsl:38-132, CPU models test/utils.rs:222-338).
//
// This stage is bound by the DEPTH of dependent group additions (a lone wave needs ~7.5 us per addition), not by bytes
// or work.  The reference's running sums (128 buckets per thread, then a 15-bit double-and-add, bpr.template.wgsl:66-75,
// 124-125) are ~270 additions deep; shortening the runs only trades depth for double-and-add work.  Instead the weighted
// sum is decomposed into PLAIN sums, which are shallow trees.  Slot 0 carries weight h = 2^15, so bucket position q
// (1..32768) reads slot q & 32767; write q - 1 = 128 * hi + lo:
//     S = sum_q q B_q = 128 * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo,   R_hi = sum_lo B[hi][lo],  C_lo = sum_hi B[hi][lo]
//   k_bpr_rowcol  the 256 row sums and 128 column sums of every window: 3 serial additions + a 5- or 6-level LDS tree
//   k_bpr_w256    W(X) = sum_{i<256} i * X_i by the same split applied twice more (16 x 16, then 4 x 4): ~16 additions deep
//   k_bpr_final   S = 128 * W(R) + W(C) + sum(C): 7 doublings + 2 additions, emits the window sum as canonical bytes
// Work: 2 additions per bucket (the minimum of the running-sum scheme) + O(1) per window; depth ~33 additions.
constexpr int BPR_ROWS = 256, BPR_COLS = 128;

// tree-add `count` (power of two) records spaced `stride` records apart starting at x[base]; every thread of the block
// must call it (it contains barriers); on return x[base] holds the sum.  `id` enumerates jobs block-wide.

// LOG_R = log2 of the buckets each thread adds serially before the LDS tree.  The host picks 4 (16 buckets) when many
// windows are reduced at once -- fewer, better-filled wave-additions: the stage is then bound by the ~7 us a SIMD needs
// per wave-addition -- and 2 for few windows, where only the depth counts.
// LOG_ROWS = log2 of the rows of the window's bucket grid: 2^(C-1) buckets = 2^LOG_ROWS rows x 128 columns (8 / 6 / 4 for
// C = 16 / 14 / 12).  The row sums of window w land in rows[w][0 .. ROWS) (stride BPR_ROWS), the column sums in cols[w][0 .. 128).
template <int LOG_R, int LOG_ROWS>
__global__ void __launch_bounds__(256, REDUCE_WAVES_PER_SIMD) k_bpr_rowcol(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ rows,
                                                    uint32_t* __restrict__ cols) {
  constexpr int R = 1 << LOG_R, ROWS = 1 << LOG_ROWS, NB = ROWS * BPR_COLS;
  static_assert(ROWS <= BPR_ROWS && R <= ROWS && R <= BPR_COLS, "bucket grid");
  constexpr int ROW_LANES = BPR_COLS / R, COL_LANES = ROWS / R;  // threads per row / per column
  constexpr int ROWS_PER_BLOCK = 256 / ROW_LANES, COLS_PER_BLOCK = 256 / COL_LANES;
  constexpr int ROW_BLOCKS = (ROWS + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  __shared__ uint32_t x[256 * XYZZ_WORDS];
  const int w = blockIdx.y, t = threadIdx.x;
  const uint32_t* bw = buckets + (size_t)w * NB * REC_WORDS;
  const bool row_block = (int)blockIdx.x < ROW_BLOCKS;
  // element e = q - 1 = 128 * hi + lo  ->  slot (e + 1) & (NB - 1)
  int e0, estep, group, lanes_per_group, out_index;
  bool live = true;  // a block may hold more row / column groups than the grid has left (small grids)
  if (row_block) {  // R consecutive lo per thread
    group = t / ROW_LANES;
    const int row = blockIdx.x * ROWS_PER_BLOCK + group, seg = t % ROW_LANES;
    live = row < ROWS;
    e0 = row * BPR_COLS + seg * R;
    estep = 1;
    lanes_per_group = ROW_LANES;
    out_index = row;
  } else {  // R consecutive hi per thread
    group = t / COL_LANES;
    const int col = ((int)blockIdx.x - ROW_BLOCKS) * COLS_PER_BLOCK + group, part = t % COL_LANES;
    live = col < BPR_COLS;
    e0 = part * R * BPR_COLS + col;
    estep = BPR_COLS;
    lanes_per_group = COL_LANES;
    out_index = col;
  }
  g1_xyzz acc = g1_identity();
  if (live) {
    acc = ld_rec(bw + (size_t)((e0 + 1) & (NB - 1)) * REC_WORDS);
#pragma unroll 1
    for (int i = 1; i < R; i++) acc = g1_add(acc, ld_rec(bw + (size_t)((e0 + i * estep + 1) & (NB - 1)) * REC_WORDS));
  }
  st_xyzz(x + t * XYZZ_WORDS, acc);
  __syncthreads();
  const int k = t & (lanes_per_group - 1);
  for (int sft = lanes_per_group >> 1; sft >= 1; sft >>= 1) {
    if (k < sft) lds_add_pair(x, t, t + sft);
    __syncthreads();
  }
  if (k == 0 && live) {
    uint32_t* out = (row_block ? rows + (size_t)w * BPR_ROWS * XYZZ_WORDS : cols + (size_t)w * 256 * XYZZ_WORDS) + (size_t)out_index * XYZZ_WORDS;
    for (int i = 0; i < XYZZ_WORDS; i++) out[i] = x[t * XYZZ_WORDS + i];
  }
}
template <int LOG_R, int LOG_ROWS>
constexpr int bpr_rowcol_blocks() {
  constexpr int R = 1 << LOG_R, ROWS = 1 << LOG_ROWS;
  constexpr int rpb = 256 / (BPR_COLS / R), cpb = 256 / (ROWS / R);
  return (ROWS + rpb - 1) / rpb + (BPR_COLS + cpb - 1) / cpb;
}

// ---- cooperative group operations: 8 lanes share ONE addition / doubling -------------------------------------------
// In the narrow tail of the reduction only a few additions are independent, so most lanes of a wave idle while a lone wave
// needs ~12 us per XYZZ addition (14 dependent field multiplications).  The multiplications of one addition are mostly
// independent of each other (critical path 4), so an octet of lanes computes them side by side: lane role r = lane & 7
// takes one product per stage, operands and intermediate results travel through LDS, stages are separated by block
// barriers.  An addition then costs 4 multiplication latencies + 5 barriers, a doubling 3 + 4.
// Every thread of the block must call these functions (they contain __syncthreads); octet o = threadIdx.x >> 3 works on
// its own operands `pa`, `pb` -> `pout` (LDS pointers to XYZZ_WORDS records; pout may alias pa or pb) when `active`.
constexpr int COOP_WORDS = 16 * FQ_L;  // scratch words per octet

__device__ __forceinline__ fq ldf(const uint32_t* p) {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_L; i++) r.v[i] = p[i];
  return r;
}
__device__ __forceinline__ void stf(uint32_t* p, const fq& a) {
#pragma unroll
  for (int i = 0; i < FQ_L; i++) p[i] = a.v[i];
}
__device__ __forceinline__ void coop_copy(uint32_t* dst, const uint32_t* src, int r) {  // 8 lanes copy one record
  if (dst != src)
    for (int i = r; i < XYZZ_WORDS; i += 8) dst[i] = src[i];
}

__device__ __noinline__ void coop_add(uint32_t* sc_all, const uint32_t* pa, const uint32_t* pb, uint32_t* pout, bool active) {
  const int r = threadIdx.x & 7;
  uint32_t* sc = sc_all + (threadIdx.x >> 3) * COOP_WORDS;
  // mode 0: full addition; 1: result = a (b is the identity); 2: result = b; 3: inactive
  int mode = 3;
  if (active) mode = pa[4 * FQ_L] != 0 ? 2 : (pb[4 * FQ_L] != 0 ? 1 : 0);
  fq keep = fq_zero();  // r0 keeps P, r1 keeps R across stages
  // stage 1: U1 = ax*bzz, U2 = bx*azz, S1 = ay*bzzz, S2 = by*azzz, ZZ12 = azz*bzz, ZZZ12 = azzz*bzzz  -> sc[0..5]
  if (mode == 0 && r < 6) {
    const uint32_t *fa, *fb;
    if (r < 4) {
      const uint32_t* first = (r & 1) ? pb : pa;
      const uint32_t* second = (r & 1) ? pa : pb;
      fa = first + ((r & 2) ? FQ_L : 0);
      fb = second + ((r & 2) ? 3 * FQ_L : 2 * FQ_L);
    } else {
      fa = pa + (r == 4 ? 2 * FQ_L : 3 * FQ_L);
      fb = pb + (r == 4 ? 2 * FQ_L : 3 * FQ_L);
    }
    stf(sc + r * FQ_L, fq_mul(ldf(fa), ldf(fb)));
  }
  __syncthreads();
  // stage 2: r0: P = U2 - U1, PP = P^2 -> sc[6] ; r1: R = S2 - S1 -> sc[9], RR = R^2 -> sc[7]
  if (mode == 0 && r < 2) {
    keep = fq_sub<3>(ldf(sc + (r == 0 ? 1 : 3) * FQ_L), ldf(sc + (r == 0 ? 0 : 2) * FQ_L));
    stf(sc + (6 + r) * FQ_L, fq_sqr(keep));
    if (r == 1) stf(sc + 9 * FQ_L, keep);
  }
  __syncthreads();
  bool special = false;  // equal x coordinates: doubling or cancellation, done serially by lane 0 at the end
  if (mode == 0) special = fq_is_zero_exact(ldf(sc + 6 * FQ_L));
  // stage 3: PPP = P*PP -> sc[10], Q = U1*PP -> sc[11], ZZ3 = ZZ12*PP -> sc[12]
  if (mode == 0 && !special && r < 3) {
    const fq PP = ldf(sc + 6 * FQ_L);
    const fq other = r == 0 ? keep : ldf(sc + (r == 1 ? 0 : 4) * FQ_L);
    stf(sc + (10 + r) * FQ_L, fq_mul(other, PP));
  }
  __syncthreads();
  // stage 4: r0: X3 = RR - PPP - 2Q -> sc[13], Y3 = R*(Q - X3) - S1*PPP -> sc[14] ; r1: ZZZ3 = ZZZ12*PPP -> sc[15]
  if (mode == 0 && !special && r < 2) {
    const fq PPP = ldf(sc + 10 * FQ_L);
    if (r == 0) {
      const fq Q = ldf(sc + 11 * FQ_L);
      const fq X3 = fq_sub<7>(ldf(sc + 7 * FQ_L), fq_add(PPP, fq_dbl(Q)));
      const fq T = fq_sub<10>(Q, X3);
      const fq nS1 = fq_sub<3>(fq_zero(), ldf(sc + 2 * FQ_L));
      stf(sc + 13 * FQ_L, X3);
      stf(sc + 14 * FQ_L, fq_mul2(ldf(sc + 9 * FQ_L), T, nS1, PPP));
    } else {
      stf(sc + 15 * FQ_L, fq_mul(ldf(sc + 5 * FQ_L), PPP));
    }
  }
  __syncthreads();
  if (mode == 0) {
    if (special) {
      if (r == 0) st_xyzz(pout, g1_add(ld_xyzz(pa), ld_xyzz(pb)));
    } else if (r < 4) {
      const int src = r == 0 ? 13 : (r == 1 ? 14 : (r == 2 ? 12 : 15));
      stf(pout + r * FQ_L, ldf(sc + src * FQ_L));
    } else if (r == 4) {
      pout[4 * FQ_L] = 0;
    }
  } else if (mode == 1) {
    coop_copy(pout, pa, r);
  } else if (mode == 2) {
    coop_copy(pout, pb, r);
  }
  __syncthreads();
}

__device__ __noinline__ void coop_double(uint32_t* sc_all, const uint32_t* pa, uint32_t* pout, bool active) {
  const int r = threadIdx.x & 7;
  uint32_t* sc = sc_all + (threadIdx.x >> 3) * COOP_WORDS;
  const bool work = active && pa[4 * FQ_L] == 0;  // doubling the identity leaves it unchanged
  fq keep = fq_zero();                      // r0 keeps U = 2Y
  // stage 1: r0: V = U^2 -> sc[0] ; r1: XX = X^2 -> sc[1]
  if (work && r < 2) {
    if (r == 0) {
      keep = fq_dbl(ldf(pa + FQ_L));
      stf(sc + 0 * FQ_L, fq_sqr(keep));
    } else {
      stf(sc + 1 * FQ_L, fq_sqr(ldf(pa)));
    }
  }
  __syncthreads();
  // stage 2: r0: W = U*V -> sc[2] ; r1: S = X*V -> sc[3] ; r2: M = 3*XX -> sc[5], MM = M^2 -> sc[4] ; r3: ZZ3 = V*ZZ -> sc[6]
  if (work && r < 4) {
    if (r == 2) {
      const fq XX = ldf(sc + 1 * FQ_L);
      const fq M = fq_norm(fq_add(fq_dbl(XX), XX));
      stf(sc + 5 * FQ_L, M);
      stf(sc + 4 * FQ_L, fq_sqr(M));
    } else {
      const fq V = ldf(sc + 0 * FQ_L);
      const fq other = r == 0 ? keep : ldf(pa + (r == 1 ? 0 : 2 * FQ_L));
      stf(sc + (r == 0 ? 2 : (r == 1 ? 3 : 6)) * FQ_L, fq_mul(other, V));
    }
  }
  __syncthreads();
  // stage 3: r0: X3 = MM - 2S -> sc[7], Y3 = M*(S - X3) - W*Y -> sc[8] ; r1: ZZZ3 = W*ZZZ -> sc[9]
  if (work && r < 2) {
    const fq W = ldf(sc + 2 * FQ_L);
    if (r == 0) {
      const fq S = ldf(sc + 3 * FQ_L);
      const fq X3 = fq_sub<5>(ldf(sc + 4 * FQ_L), fq_dbl(S));
      const fq T = fq_sub<8>(S, X3);
      const fq nY = fq_sub<6>(fq_zero(), ldf(pa + FQ_L));
      stf(sc + 7 * FQ_L, X3);
      stf(sc + 8 * FQ_L, fq_mul2(ldf(sc + 5 * FQ_L), T, nY, W));
    } else {
      stf(sc + 9 * FQ_L, fq_mul(W, ldf(pa + 3 * FQ_L)));
    }
  }
  __syncthreads();
  if (work) {
    if (r < 4) {
      const int src = r == 0 ? 7 : (r == 1 ? 8 : (r == 2 ? 6 : 9));
      stf(pout + r * FQ_L, ldf(sc + src * FQ_L));
    } else if (r == 4) {
      pout[4 * FQ_L] = 0;
    }
  } else if (active) {
    coop_copy(pout, pa, r);
  }
  __syncthreads();
}

// W(X) = sum_{i<256} i * X_i for X = rows (blockIdx.x == 0) or the columns padded to 256 with identities (blockIdx.x == 1);
// out[w][blockIdx.x] = W(X); for the columns additionally out[w][2] = sum(X).
// 16 x 16 split (RR_a = row sums, CC_b = column sums of the 16 x 16 arrangement of X): W = 16 * W16(RR) + W16(CC);
// W16(V) by a 4 x 4 split: W16 = 4 * W4(r) + W4(c); W4(u) = u1 + 2 u2 + 3 u3 = (u1 + u3) + 2 (u2 + u3).
// Levels with at most 32 independent operations use the cooperative octet operations above.
// (needs 256 records + 32 octets of scratch in LDS: 50 KB with 9 limbs; with 14 limbs it would pass the 64 KB a workgroup may declare, so
//  a unit of that size finishes its window sums from the bit-plane sums instead: k_bpr_planes<true> + k_bpr_final_planes below)
constexpr bool BPR_USE_W256 = (256 * XYZZ_WORDS + 32 * COOP_WORDS) * 4 <= 65536;
__global__ void __launch_bounds__(256) k_bpr_w256(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ cols,
                                                  uint32_t* __restrict__ out, int nrows) {
  __shared__ uint32_t x[BPR_USE_W256 ? 256 * XYZZ_WORDS : 1];
  __shared__ uint32_t sc[BPR_USE_W256 ? 32 * COOP_WORDS : 1];
  if constexpr (!BPR_USE_W256) return;
  const int w = blockIdx.y, which_in = blockIdx.x, t = threadIdx.x;
  const int a = t >> 4, b = t & 15, o = t >> 3;  // o: octet index, 0..31
  auto X = [&](int i) { return x + i * XYZZ_WORDS; };
  // after the trees only records 8g of x stay live; the free ones hold the small intermediate vectors
  auto Y = [&](int j) { return x + (8 * (j >> 1) + 1 + (j & 1)) * XYZZ_WORDS; };  // j < 64
  auto Z = [&](int j) { return x + (8 * j + 3) * XYZZ_WORDS; };                   // j < 16
  auto U = [&](int j) { return x + (8 * j + 4) * XYZZ_WORDS; };                   // j < 8
  auto T = [&](int j) { return x + (8 * j + 5) * XYZZ_WORDS; };                   // j < 8 (total of the row sums)
  auto Q = [&](int j) { return x + (8 * j + 6) * XYZZ_WORDS; };                   // j < 8 (W4 temporaries)
  g1_xyzz xi;
  if (which_in == 0) xi = t < nrows ? ld_xyzz(rows + ((size_t)w * BPR_ROWS + t) * XYZZ_WORDS) : g1_identity();  // rows padded to 256
  else xi = t < BPR_COLS ? ld_xyzz(cols + ((size_t)w * 256 + t) * XYZZ_WORDS) : g1_identity();
  st_xyzz(X(t), xi);
  __syncthreads();
  // level 1 (256 additions, all lanes busy): threads with b < 8 add row pairs, the others add column pairs
  g1_xyzz l1;
  int l1_dst;
  if (b < 8) {
    l1 = g1_add(xi, ld_xyzz(X(16 * a + b + 8)));
    l1_dst = a * 8 + b;  // row partials: [0, 128)
  } else {
    const int job = a * 8 + (b - 8), c = job & 15, pr = job >> 4;
    l1 = g1_add(ld_xyzz(X(16 * pr + c)), ld_xyzz(X(16 * (pr + 8) + c)));
    l1_dst = 128 + c * 8 + pr;  // column partials: [128, 256)
  }
  __syncthreads();
  st_xyzz(X(l1_dst), l1);
  __syncthreads();
  for (int sft = 4; sft >= 2; sft >>= 1) {  // 32 groups of 8 -> 2 (128 and 64 additions)
    if (t < 32 * sft) {
      const int g = t / sft, k = t % sft;
      lds_add_pair(x, g * 8 + k, g * 8 + k + sft);
    }
    __syncthreads();
  }
  coop_add(sc, X(o * 8), X(o * 8 + 1), X(o * 8), true);  // 32 groups: 2 -> 1
  // V0[i] = X(8 i) (16 row sums RR), V1[i] = X(128 + 8 i) (16 column sums CC)
  auto V = [&](int v, int i) { return X(v * 128 + 8 * i); };
  {  // 4 x 4 split, level 1: 32 jobs -> Y(v*16 + job)
    const int v = o >> 4, job = o & 15;
    const uint32_t *pa, *pb;
    if (job < 8) {  // row pair (i, j): V[4i + j] + V[4i + j + 2]
      const int i = job >> 1, j = job & 1;
      pa = V(v, 4 * i + j);
      pb = V(v, 4 * i + j + 2);
    } else {  // column pair (j, pr): V[4 pr + j] + V[4 (pr + 2) + j]
      const int j = (job - 8) >> 1, pr = job & 1;
      pa = V(v, 4 * pr + j);
      pb = V(v, 4 * (pr + 2) + j);
    }
    coop_add(sc, pa, pb, Y(v * 16 + job), true);
  }
  {  // level 2: r_i, c_j (16 jobs) -> Z(v*8 + q) ; total of V0, level 1 (8 jobs) -> T(k)
    const uint32_t *pa = x, *pb = x;
    uint32_t* po = x;
    const bool act = o < 24;
    if (o < 16) {
      const int v = o >> 3, q = o & 7;  // q < 4: r_q ; q >= 4: c_{q-4}
      const int base = v * 16 + (q < 4 ? 2 * q : 8 + 2 * (q - 4));
      pa = Y(base);
      pb = Y(base + 1);
      po = Z(v * 8 + q);
    } else if (o < 24) {
      const int k = o - 16;
      pa = V(0, k);
      pb = V(0, k + 8);
      po = T(k);
    }
    coop_add(sc, pa, pb, po, act);
  }
  {  // W4 step A: p = u1 + u3 -> U(j), q = u2 + u3 -> Q(j) for the 4 vectors j = v*2 + which ; total 8 -> 4
    const uint32_t *pa = x, *pb = x;
    uint32_t* po = x;
    const bool act = o < 12;
    if (o < 8) {
      const int j = o >> 1;
      const uint32_t* q4 = Z(j * 4) - 0;  // u_i = Z(j*4 + i)
      (void)q4;
      pa = Z(j * 4 + ((o & 1) ? 2 : 1));
      pb = Z(j * 4 + 3);
      po = (o & 1) ? Q(j) : U(j);
    } else if (o < 12) {
      const int k = o - 8;
      pa = T(k);
      pb = T(k + 4);
      po = T(k);
    }
    coop_add(sc, pa, pb, po, act);
  }
  coop_double(sc, Q(o & 3), Q(o & 3), o < 4);  // W4 step B: q <- 2 q
  {  // W4 step C: W4 = p + 2q -> U(j) ; total 4 -> 2
    const uint32_t *pa = x, *pb = x;
    uint32_t* po = x;
    const bool act = o < 6;
    if (o < 4) {
      pa = U(o);
      pb = Q(o);
      po = U(o);
    } else if (o < 6) {
      const int k = o - 4;
      pa = T(k);
      pb = T(k + 2);
      po = T(k);
    }
    coop_add(sc, pa, pb, po, act);
  }
  // W16(v) = 4 * W4(r_v) + W4(c_v): U(2v) <- 4 U(2v), then U(4 + v) = U(2v) + U(2v + 1) ; total 2 -> 1
  coop_double(sc, U(2 * (o & 1)), U(2 * (o & 1)), o < 2);
  coop_double(sc, U(2 * (o & 1)), U(2 * (o & 1)), o < 2);
  {
    const uint32_t *pa = x, *pb = x;
    uint32_t* po = x;
    const bool act = o < 3;
    if (o < 2) {
      pa = U(2 * o);
      pb = U(2 * o + 1);
      po = U(4 + o);
    } else if (o == 2) {
      pa = T(0);
      pb = T(1);
      po = T(0);
    }
    coop_add(sc, pa, pb, po, act);
  }
  // W256 = 16 * W16(RR) + W16(CC)
  for (int i = 0; i < 4; i++) coop_double(sc, U(4), U(4), o == 0);
  coop_add(sc, U(4), U(5), U(4), o == 0);
  if (t < XYZZ_WORDS) out[((size_t)w * 3 + which_in) * XYZZ_WORDS + t] = U(4)[t];
  if (which_in == 1 && t >= 64 && t < 64 + XYZZ_WORDS) out[((size_t)w * 3 + 2) * XYZZ_WORDS + (t - 64)] = T(0)[t - 64];
}

// one lane per window: S = 128 * W(R) + W(C) + sum(C), emitted as canonical Jacobian bytes.  (Operands stay in registers
// here, which measured faster than the cooperative LDS form: 52 vs 71 us.)
// (every kernel that ends a launch's reduce chain also hands the launch's error word to the host -- err_host is the slot's pinned word, written
//  directly -- and clears it for the slot's next occupant: two copies and a fill less at the end of every chain)
__device__ __forceinline__ void finish_error_word(uint32_t* __restrict__ err_dev, uint32_t* __restrict__ err_host) {
  *err_host = *err_dev;
  *err_dev = 0;
}
// (emit_total: shares of the wide tables' virtual windows -- record 2 w is the window sum, record 2 w + 1 the window's PLAIN total
//  TC_w = sum_slot B[w][slot], which the finish needs beside it: host_g1.h, combine_wide_pairs)
__global__ void __launch_bounds__(64) k_bpr_final(const uint32_t* __restrict__ parts, int w_count, uint32_t* __restrict__ wsums,
                                                  uint32_t* __restrict__ big_queue, uint32_t* __restrict__ err_dev, uint32_t* __restrict__ err_host,
                                                  int emit_total) {
  const int w = threadIdx.x;
  if (w == 0) {
    big_queue[0] = 0;  // the stitch's queue of big buckets, consumed earlier on this stream: empty for the next launch
    finish_error_word(err_dev, err_host);
  }
  if (w >= w_count) return;
  g1_xyzz acc = ld_xyzz(parts + ((size_t)w * 3 + 0) * XYZZ_WORDS);
  for (int i = 0; i < 7; i++) acc = g1_double(acc);
  const g1_xyzz total = ld_xyzz(parts + ((size_t)w * 3 + 2) * XYZZ_WORDS);
  acc = g1_add(acc, g1_add(ld_xyzz(parts + ((size_t)w * 3 + 1) * XYZZ_WORDS), total));
  st_jacobian_plain(wsums + (size_t)(emit_total ? 2 * w : w) * JAC_WORDS, acc);
  if (emit_total) st_jacobian_plain(wsums + (size_t)(2 * w + 1) * JAC_WORDS, total);
}

// ... or, for a launch whose sums go to the host anyway (one MSM per launch: its LATENCY is what counts): the narrow end of the reduction
// is not done here at all.  W(X) = sum_i i X_i = sum_b 2^b P_b with the PLAIN bit-plane sums P_b = sum_{i: bit b of i set} X_i -- eight
// (rows) + seven (columns) + the column total = PLANES_PER_WINDOW masked tree sums per window, all independent and 8 additions deep,
// instead of k_bpr_w256's ~16 dependent levels and k_bpr_final's 9 operations in a lone wave (~200 us together); the positional
// combination S_w = sum_b 2^(b+7) PR_b + sum_b 2^b PC_b + TC is 29 group operations per window on the host (host_g1.h:
// window_sum_from_planes, ~8 us; the windows side by side on the host pool).
// Grid (PLANES_PER_WINDOW, windows); plane 0 .. 7: row bit b, 8 .. 14: column bit b - 8, 15: column total.  out[w][plane] x 96 B Jacobian.
// XYZZ_OUT: the plane sums stay on the device as XYZZ records (k_bpr_final_planes finishes the window sums there).
constexpr int PLANES_PER_WINDOW = 16;
template <bool XYZZ_OUT>
__global__ void __launch_bounds__(256) k_bpr_planes(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ cols, uint32_t* __restrict__ out,
                                                    int nrows, uint32_t* __restrict__ big_queue, uint32_t* __restrict__ err_dev,
                                                    uint32_t* __restrict__ err_host) {
  __shared__ uint32_t x[256 * XYZZ_WORDS];
  const int w = blockIdx.y, plane = blockIdx.x, t = threadIdx.x;
  if (!XYZZ_OUT && w == 0 && plane == 0 && t == 0) {  // as k_bpr_final: this kernel ends the chain (`out` may be the slot's pinned host buffer itself)
    big_queue[0] = 0;
    finish_error_word(err_dev, err_host);
  }
  const bool is_row = plane < 8;
  const int bit = is_row ? plane : plane - 8;  // 7 for the column total: bit 7 of a column index is never set -> handled by `all`
  const bool all = plane == PLANES_PER_WINDOW - 1;
  const int count = is_row ? nrows : BPR_COLS;
  const bool take = t < count && (all || ((t >> bit) & 1));
  const uint32_t* src = is_row ? rows + ((size_t)w * BPR_ROWS + t) * XYZZ_WORDS : cols + ((size_t)w * 256 + t) * XYZZ_WORDS;
  // first level straight from memory: thread t < 128 adds elements t and t + 128 (both masked)
  g1_xyzz acc = g1_identity();
  if (t < 128) {
    if (take) acc = ld_xyzz(src);
    const int u = t + 128;
    if (u < count && (all || ((u >> bit) & 1))) acc = g1_add(acc, ld_xyzz(src + (size_t)128 * XYZZ_WORDS));
    st_xyzz(x + t * XYZZ_WORDS, acc);
  }
  __syncthreads();
  for (int sft = 64; sft >= 1; sft >>= 1) {
    if (t < sft) lds_add_pair(x, t, t + sft);
    __syncthreads();
  }
  if constexpr (XYZZ_OUT) {
    if (t < XYZZ_WORDS) out[((size_t)w * PLANES_PER_WINDOW + plane) * XYZZ_WORDS + t] = x[t];
  } else {
    if (t == 0) st_jacobian_plain(out + ((size_t)w * PLANES_PER_WINDOW + plane) * JAC_WORDS, ld_xyzz(x));
  }
}
// one lane per window: S_w = sum_b 2^(b+7) PR_b + sum_b 2^b PC_b + TC from the plane sums (XYZZ records), as canonical Jacobian bytes --
// the device-side counterpart of host_g1.h: window_sum_from_planes, for sums that stay on the device
__global__ void __launch_bounds__(64) k_bpr_final_planes(const uint32_t* __restrict__ planes, int w_count, uint32_t* __restrict__ wsums,
                                                         uint32_t* __restrict__ big_queue, uint32_t* __restrict__ err_dev, uint32_t* __restrict__ err_host,
                                                         int emit_total) {
  const int w = threadIdx.x;
  if (w == 0) {
    big_queue[0] = 0;
    finish_error_word(err_dev, err_host);
  }
  if (w >= w_count) return;
  const uint32_t* pw = planes + (size_t)w * PLANES_PER_WINDOW * XYZZ_WORDS;
  g1_xyzz acc = g1_identity();
#pragma unroll 1
  for (int pos = 14; pos >= 0; pos--) {
    acc = g1_double(acc);
    acc = g1_add(acc, ld_xyzz(pw + (size_t)(pos >= 7 ? pos - 7 : 8 + pos) * XYZZ_WORDS));
  }
  const g1_xyzz total = ld_xyzz(pw + (size_t)(PLANES_PER_WINDOW - 1) * XYZZ_WORDS);
  acc = g1_add(acc, total);
  st_jacobian_plain(wsums + (size_t)(emit_total ? 2 * w : w) * JAC_WORDS, acc);
  if (emit_total) st_jacobian_plain(wsums + (size_t)(2 * w + 1) * JAC_WORDS, total);
}

// bucket records -> Jacobian wire records (stage read-back for the parity tests)
__global__ void __launch_bounds__(256) k_export_buckets(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ out, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  st_jacobian_plain(out + i * JAC_WORDS, ld_rec(buckets + i * REC_WORDS));
}

// ------------------------------------------------------------------------------------------------ samplers
// deterministic synthetic inputs (≙ sample_scalars / sample_points, src/lib.rs:20-42, but seeded)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// NW 32-bit words (8: a scalar or a coordinate of the 254 / 255-bit fields, masked to 254 bits; 12: a BLS12-381 coordinate, 381 bits)
template <int NW>
__device__ __forceinline__ void draw_words(uint64_t seed, uint64_t index, uint64_t attempt, uint64_t domain, uint32_t w[NW]) {
  uint64_t base = splitmix64(seed ^ ((domain & 0xFF) << 56)) ^ (index * 0xD1342543DE82EF95ull);
  base = splitmix64(base ^ (attempt * 0xA0761D6478BD642Full));
  uint64_t s = base;
#pragma unroll
  for (int i = 0; i < NW / 2; i++) {
    s = splitmix64(s);
    w[2 * i] = (uint32_t)s;
    w[2 * i + 1] = (uint32_t)(s >> 32);
  }
  if constexpr (NW == 8) w[7] &= 0x3FFFFFFFu;        // 254 bits
  else w[NW - 1] &= (1u << (FQ_BITS - 32 * (NW - 1))) - 1u;  // as many bits as p has
}
__device__ __forceinline__ void draw256(uint64_t seed, uint64_t index, uint64_t attempt, uint64_t domain, uint32_t w[8]) {
  draw_words<8>(seed, index, attempt, domain, w);
}

__global__ void __launch_bounds__(256) k_sample_scalars(uint64_t seed, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  for (uint64_t attempt = 0;; attempt++) {
    draw256(seed, i, attempt, 1, w);
    if (!geq_modulus<1>(w)) break;
  }
  st8(out + i * 8, w);
}

#ifndef MSM_FQ2
__device__ __constant__ cwords<CW> c_sqrt_t = make_cwords(FQ_SQRT_T_32);
__device__ __constant__ cwords<CW> c_sqrt_tp1h = make_cwords(FQ_SQRT_TP1H_32);
__device__ __forceinline__ fq fq_pow254(const fq& a, const uint32_t* e) {  // a^e, e < 2^(32 CW - 2) (constant memory), a exact
  fq acc = fq_one();
  for (int bit = 32 * CW - 3; bit >= 0; bit--) {
    acc = fq_sqr(acc);
    if ((e[bit >> 5] >> (bit & 31)) & 1u) acc = fq_mul(acc, a);
  }
  return acc;
}
// a candidate square root of a (the caller checks y^2 == a).  p = 3 mod 4 (BN254 Fq): a^((p+1)/4).  Otherwise (Grumpkin's base
// field, p - 1 = 2^28 t): Tonelli-Shanks with every loop bounded by the 2-adicity, so a non-residue just yields a wrong candidate.
__device__ __forceinline__ fq fq_sqrt_candidate(const fq& a) {  // a exact
  if constexpr (FQ_SQRT_S == 0) {
    return fq_pow254(a, c_pp1d4.w);
  } else {
    fq x = fq_pow254(a, c_sqrt_tp1h.w), b = fq_pow254(a, c_sqrt_t.w), c;
#pragma unroll
    for (int i = 0; i < FQ_L; i++) c.v[i] = FQ_SQRT_C0_29[i];
    const fq one = fq_canonical(fq_one());
    int m = FQ_SQRT_S;
    for (int round = 0; round < FQ_SQRT_S; round++) {
      if (fq_equal_exact(fq_canonical(b), one)) break;
      int i = 0;  // least i with b^(2^i) == 1
      fq q = b;
      while (i < m && !fq_equal_exact(fq_canonical(q), one)) {
        q = fq_sqr(q);
        i++;
      }
      if (i >= m) break;  // not a square
      fq bb = c;
      for (int k = 0; k < m - i - 1; k++) bb = fq_sqr(bb);
      x = fq_mul(x, bb);
      c = fq_sqr(bb);
      b = fq_mul(b, c);
      m = i;
    }
    return x;
  }
}

__global__ void __launch_bounds__(256) k_sample_points(uint64_t seed, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[CW];
  for (uint64_t attempt = 0;; attempt++) {
    draw_words<CW>(seed, i, attempt, 2, wx);
    if (geq_modulus<0>(wx)) continue;
    const fq x = fq_to_mont(fq_unpack(wx));
    const fq rhs = fq_canonical(fq_tidy(fq_add(fq_mul(fq_sqr(x), x), fq_curve_b())));
    const fq y = fq_sqrt_candidate(rhs);
    if (!fq_equal_exact(fq_canonical(fq_sqr(y)), rhs)) continue;
    fq yp = fq_from_mont(y);  // canonical integer
    if ((yp.v[0] & 1u) != ((wx[0] >> 1) & 1u)) yp = fq_neg_canonical(yp);
    st_coord(out + i * PT_WORDS, wx);
    st_fq(out + i * PT_WORDS + CW, yp);
    break;
  }
}
#else  // MSM_FQ2
// G2: P_i = (a + i b) G for seeded odd a, b < r and the standard generator G of the order-r subgroup -- byte for byte what the oracle's
// sample_points(n, seed) produces (oracle/bn254_g2_ref.py; its sample_multipliers gives every MSM over these points a closed form).  Points
// of G2 proper, unlike try-and-increment on the twist (whose cofactor is huge): fit for the endomorphism mode.  One lane per point: a
// double-and-add over the 288-bit integer a + i b (not reduced: G has order r), then one inversion in Fq2.
__global__ void __launch_bounds__(256) k_sample_points(uint64_t seed, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t ab[2][8];
  for (int which = 0; which < 2; which++) {  // a, b = sample_scalar(seed ^ 0x6732, 0 / 1) | 1
    for (uint64_t attempt = 0;; attempt++) {
      draw256(seed ^ 0x6732ull, (uint64_t)which, attempt, 1, ab[which]);
      if (!geq_modulus<1>(ab[which])) break;
    }
    ab[which][0] |= 1u;
  }
  uint32_t m[10];  // a + i b < 2^254 + 2^28 2^254
  {
    const uint64_t lo = (uint32_t)i, hi = (uint64_t)i >> 32;  // (i < 2^28: hi = 0; kept general)
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
      uint64_t t = c + (k < 8 ? ab[0][k] : 0u);
      uint64_t carry = t >> 32;
      t &= 0xffffffffull;
      if (k < 8) {
        const uint64_t p0 = lo * ab[1][k];
        t += p0 & 0xffffffffull;
        carry += p0 >> 32;
      }
      if (k >= 1 && k < 9) {
        const uint64_t p1 = hi * ab[1][k - 1];
        t += p1 & 0xffffffffull;
        carry += p1 >> 32;
      }
      m[k] = (uint32_t)t;
      c = carry + (t >> 32);
    }
  }
  fq gx, gy;
#pragma unroll
  for (int k = 0; k < FQ_L; k++) {
    gx.v[k] = FQ_GEN_X29[k];
    gy.v[k] = FQ_GEN_Y29[k];
  }
  const g1_xyzz g = g1_from_affine(gx, gy);
  g1_xyzz acc = g1_identity();
#pragma unroll 1
  for (int bit = 32 * 10 - 1; bit >= 0; bit--) {
    acc = g1_double(acc);
    if ((m[bit >> 5] >> (bit & 31)) & 1u) acc = g1_add(acc, g);
  }
  // a + i b is not a multiple of r for the sizes that fit a context (a, b odd, i < 2^28): acc is a point
  const fq t = fq_inv(fq_mul(acc.zz, acc.zzz));
  st_fq(out + i * PT_WORDS, fq_from_mont(fq_mul(acc.x, fq_mul(t, acc.zzz))));
  st_fq(out + i * PT_WORDS + CW, fq_from_mont(fq_mul(acc.y, fq_mul(t, acc.zz))));
}
#endif  // MSM_FQ2

// ------------------------------------------------------------------------------------------------ op hooks for tests
// (≙ src/cuzk/wgsl/test/test_field.wgsl:13-62, test_point.wgsl:18-88)
__global__ void __launch_bounds__(256) k_test_fq(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fq x = fq_to_mont(ld_fq(a + i * CW));
  const fq y = b ? fq_to_mont(ld_fq(b + i * CW)) : fq_zero();
  fq z;
  switch (op) {
    case 0: z = fq_add(x, y); break;
    case 1: z = fq_sub<2>(x, y); break;
    case 2: z = fq_mul(x, y); break;
    case 3: z = fq_sqr(x); break;
    case 4: z = fq_neg_canonical(x); break;
    // the SMVP's multipliers (inline assembly on the device, fq29_asm.h), called directly; 8 and 9 feed them lazy limbs
    case 5: z = fq_mul_fast(x, y); break;
    case 6: z = fq_sqr_fast(x); break;
    case 7: z = fq_mul2_fast(x, y, y, x); break;                    // x y + y x
    case 8: z = fq_mul_fast(fq_add(x, y), fq_dbl(x)); break;        // (x + y) * 2x, limbs up to 2^30
    default: z = fq_sqr_fast(fq_add(x, y)); break;                  // (x + y)^2
  }
  st_fq(out + i * CW, fq_from_mont(z));
}

__global__ void __launch_bounds__(256) k_test_g1(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_xyzz p = ld_jacobian_plain(a + i * JAC_WORDS);
  g1_xyzz r;
  if (op == 0) {
    r = g1_add(p, ld_jacobian_plain(b + i * JAC_WORDS));
  } else if (op == 1) {
    r = g1_double(p);
  } else if (op == 2) {
    const fq qx = fq_to_mont(ld_fq(b + i * PT_WORDS)), qy = fq_to_mont(ld_fq(b + i * PT_WORDS + CW));
    g1_madd(p, qx, qy);
    r = p;
  } else {  // the SMVP's signed-state form (g1_madd_w): 3: p + q - q + q ; 4: p - q - q  (every sign state, both digit signs)
    const fq qx = fq_to_mont(ld_fq(b + i * PT_WORDS)), qy = fq_to_mont(ld_fq(b + i * PT_WORDS + CW));
    bool wneg = false;
    if (op == 3) {
      g1_madd_w(p, wneg, qx, qy, false);
      g1_madd_w(p, wneg, qx, qy, true);
      g1_madd_w(p, wneg, qx, qy, false);
    } else {
      g1_madd_w(p, wneg, qx, qy, true);
      g1_madd_w(p, wneg, qx, qy, true);
    }
    r = g1_unsigned(p, wneg);
  }
  st_jacobian_plain(out + i * JAC_WORDS, r);
}

__global__ void __launch_bounds__(256) k_test_g1_mul_u32(const uint32_t* __restrict__ a, const uint32_t* __restrict__ k,
                                                         uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_jacobian_plain(out + i * JAC_WORDS, g1_mul_u32(ld_jacobian_plain(a + i * JAC_WORDS), k[i]));
}

}  // namespace MSM_KERNEL_NS
