// Device kernels of the cuZK-style BN254 MSM pipeline for gfx950 (MI355X).  Included once by msm_hip.hip.
//
// Pipeline (reference: compute_msm, src/cuzk/msm.rs:75-417) and the HBM layout each stage reads/writes:
//
//   bases     u32[n][16]            packed affine, Montgomery (R = 2^261), x || y, 64 B per point, resident
//   scalars   u32[n][8]             canonical little-endian (wire format)
//   digits    u16[W][n]             signed-magnitude digit codes, window-major planes
//   col_ptr   u32[W][32769]         start of every bucket slot in val_idxs (exclusive scan of the histogram)
//   val_idxs  u32[W][n]             point index | sign << 31, grouped by bucket slot
//   buckets   u32[W][32768][24]     Jacobian, Montgomery, 96 B per bucket
//   wsums     u8 [W][96]            window sums, Jacobian, canonical little-endian (leaves the device)
//
// The reference keys its CSC rows by the biased digit (65536 rows per window, transpose.template.wgsl:47-73) and lets
// the SMVP thread visit rows h+k and h-k (smvp.template.wgsl:55-92).  Here the sort key is the bucket slot itself
// (|d| mod 2^15, 32768 rows) and the sign rides in bit 31 of the index, so one bucket is one contiguous run.
#pragma once
#include <hip/hip_runtime.h>

#include "g1.h"

namespace msmk {
using namespace bn254;

constexpr int WBITS = 16;
constexpr int NWIN = 16;
constexpr int HALF = 1 << (WBITS - 1);  // 32768 bucket slots per window

__device__ __constant__ uint32_t c_pp1d4[8] = {FQ_PP1D4_32[0], FQ_PP1D4_32[1], FQ_PP1D4_32[2], FQ_PP1D4_32[3],
                                               FQ_PP1D4_32[4], FQ_PP1D4_32[5], FQ_PP1D4_32[6], FQ_PP1D4_32[7]};

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
__device__ __forceinline__ fq ld_fq(const uint32_t* p) {  // packed -> limbs (no domain change)
  uint32_t w[8];
  ld8(p, w);
  return fq_unpack(w);
}
__device__ __forceinline__ void st_fq(uint32_t* p, const fq& x) {  // x exact, < 2^256
  uint32_t w[8];
  fq_pack(w, x);
  st8(p, w);
}
// w >= modulus ?   MOD = 0: Fq modulus p, MOD = 1: Fr modulus r  (constants fold to immediates)
template <int MOD>
__device__ __forceinline__ bool geq_modulus(const uint32_t w[8]) {
  bool gt = false, lt = false;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    const uint32_t m = MOD == 0 ? FQ_P32[i] : FR_R32[i];
    gt = gt || (!lt && w[i] > m);
    lt = lt || (!gt && w[i] < m);
  }
  return !lt;
}
__device__ __forceinline__ bool fq_equal_exact(const fq& a, const fq& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
__device__ __forceinline__ fq fq_three() {
  fq r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = FQ_THREE29[i];
  return r;
}

// Jacobian record (24 words, Montgomery canonical) <-> XYZZ registers
__device__ __forceinline__ g1_xyzz ld_jacobian(const uint32_t* p) {
  const fq X = ld_fq(p), Y = ld_fq(p + 8), Z = ld_fq(p + 16);
  return g1_from_jacobian(X, Y, Z);
}
__device__ __forceinline__ void st_jacobian(uint32_t* p, const g1_xyzz& a) {
  fq X, Y, Z;
  g1_to_jacobian(a, X, Y, Z);
  st_fq(p, X);
  st_fq(p + 8, Y);
  st_fq(p + 16, Z);
}
// Jacobian record, canonical non-Montgomery integers (the wire format of results)
__device__ __forceinline__ void st_jacobian_plain(uint32_t* p, const g1_xyzz& a) {
  fq X, Y, Z;
  g1_to_jacobian(a, X, Y, Z);
  st_fq(p, fq_from_mont(X));
  st_fq(p + 8, fq_from_mont(Y));
  st_fq(p + 16, fq_from_mont(Z));
}
__device__ __forceinline__ g1_xyzz ld_jacobian_plain(const uint32_t* p) {
  const fq X = fq_to_mont(ld_fq(p)), Y = fq_to_mont(ld_fq(p + 8)), Z = fq_to_mont(ld_fq(p + 16));
  return g1_from_jacobian(X, Y, Z);
}

// XYZZ record in scratch memory / LDS: 36 limbs + identity flag
constexpr int XYZZ_WORDS = 37;
template <typename PTR>
__device__ __forceinline__ void st_xyzz(PTR p, const g1_xyzz& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    p[i] = a.x.v[i];
    p[9 + i] = a.y.v[i];
    p[18 + i] = a.zz.v[i];
    p[27 + i] = a.zzz.v[i];
  }
  p[36] = a.inf ? 1u : 0u;
}
template <typename PTR>
__device__ __forceinline__ g1_xyzz ld_xyzz(PTR p) {
  g1_xyzz a;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    a.x.v[i] = p[i];
    a.y.v[i] = p[9 + i];
    a.zz.v[i] = p[18 + i];
    a.zzz.v[i] = p[27 + i];
  }
  a.inf = p[36] != 0;
  return a;
}

// error bits written to the context's device error word
constexpr uint32_t ERRBIT_NONCANONICAL = 1u;
constexpr uint32_t ERRBIT_NOT_ON_CURVE = 2u;
constexpr uint32_t ERRBIT_SCALAR_CARRY = 4u;

// ------------------------------------------------------------------------------------------------ stage 0: bases
// canonical wire bytes -> packed Montgomery affine (≙ decompose_scalars.template.wgsl:41-70, the point half)
__global__ void __launch_bounds__(256) k_convert_points(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                                                        uint32_t flags, uint32_t* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[8], wy[8];
  ld8(in + i * 16, wx);
  ld8(in + i * 16 + 8, wy);
  if (geq_modulus<0>(wx) || geq_modulus<0>(wy)) atomicOr(err, ERRBIT_NONCANONICAL);
  const fq x = fq_to_mont(fq_unpack(wx)), y = fq_to_mont(fq_unpack(wy));
  if (flags & 1u) {
    const fq lhs = fq_canonical(fq_sqr(y));
    const fq rhs = fq_canonical(fq_tidy(fq_add(fq_mul(fq_sqr(x), x), fq_three())));
    if (!fq_equal_exact(lhs, rhs)) atomicOr(err, ERRBIT_NOT_ON_CURVE);
  }
  st_fq(out + i * 16, x);
  st_fq(out + i * 16 + 8, y);
}

// ------------------------------------------------------------------------------------------------ stage 1: decompose
// 32-byte scalar -> 16 signed 16-bit digits (≙ decompose_scalars.template.wgsl:83-112, CPU model test/utils.rs:121-161):
//   d = raw + carry; if d >= 2^15 { d -= 2^16; carry = 1 }.  Stored as a signed-magnitude code
//   code = sign << 15 | (|d| & 0x7fff):  0 = digit 0 (contributes nothing), 0x8000 = digit -2^15 (bucket slot 0).
// Also builds the per-window bucket histogram (the first loop of transpose.template.wgsl:53-55).
__global__ void __launch_bounds__(256) k_decompose(const uint32_t* __restrict__ scalars, size_t n, int w_begin, int w_count,
                                                   uint16_t* __restrict__ digits, uint32_t* __restrict__ hist,
                                                   uint32_t* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  ld8(scalars + i * 8, s);
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < NWIN; w++) {
    const uint32_t raw = (s[w >> 1] >> ((w & 1) * 16)) & 0xffffu;
    uint32_t d = raw + carry;  // 0 .. 65536
    uint32_t code;
    if (d >= (uint32_t)HALF) {
      const uint32_t mag = 65536u - d;  // 0 .. 32768  (0 only when d == 65536, i.e. digit 0 with carry)
      carry = 1;
      code = mag == 0 ? 0u : (0x8000u | (mag & 0x7fffu));
    } else {
      carry = 0;
      code = d;
    }
    const int lw = w - w_begin;
    if (lw >= 0 && lw < w_count) {
      digits[(size_t)lw * n + i] = (uint16_t)code;
      if (code != 0) atomicAdd(&hist[(size_t)lw * HALF + (code & 0x7fffu)], 1u);
    }
  }
  if (carry) atomicOr(err, ERRBIT_SCALAR_CARRY);  // "final carry is 1", test/utils.rs:150-152
}

// ------------------------------------------------------------------------------------------------ stage 2a: scan
// exclusive prefix sum of the histogram of one window per block (≙ transpose.template.wgsl:58-61)
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ col_ptr,
                                               uint32_t* __restrict__ cursor) {
  __shared__ uint32_t part[1024];
  const int w = blockIdx.x, t = threadIdx.x;
  constexpr int PER = HALF / 1024;  // 32
  const uint32_t* h = hist + (size_t)w * HALF + t * PER;
  uint32_t local[PER];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k += 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(h + k);
    local[k] = v.x; local[k + 1] = v.y; local[k + 2] = v.z; local[k + 3] = v.w;
    sum += v.x + v.y + v.z + v.w;
  }
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;  // exclusive
  uint32_t* cp = col_ptr + (size_t)w * (HALF + 1) + t * PER;
  uint32_t* cu = cursor + (size_t)w * HALF + t * PER;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    cp[k] = run;
    cu[k] = run;
    run += local[k];
  }
  if (t == 1023) col_ptr[(size_t)w * (HALF + 1) + HALF] = run;
}

// ------------------------------------------------------------------------------------------------ stage 2b: scatter
// (≙ transpose.template.wgsl:66-73; order inside a slot is arrival order of the atomics, which the group sum ignores)
__global__ void __launch_bounds__(256) k_scatter(const uint16_t* __restrict__ digits, size_t n, uint32_t* __restrict__ cursor,
                                                 uint32_t* __restrict__ val_idxs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = blockIdx.y;
  if (i >= n) return;
  const uint32_t code = digits[(size_t)w * n + i];
  if (code == 0) return;
  const uint32_t pos = atomicAdd(&cursor[(size_t)w * HALF + (code & 0x7fffu)], 1u);
  val_idxs[(size_t)w * n + pos] = (uint32_t)i | ((code >> 15) << 31);
}

// ------------------------------------------------------------------------------------------------ stage 3: SMVP
// bucket accumulate, one lane per bucket slot (≙ smvp.template.wgsl:31-117, CPU model test/utils.rs:166-219):
//   B[w][k] = sum_{d=+k} P - sum_{d=-k} P  (k >= 1),   B[w][0] = -sum_{d=-2^15} P
__global__ void __launch_bounds__(256) k_smvp_bucket(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ col_ptr,
                                                     const uint32_t* __restrict__ val_idxs, size_t n,
                                                     uint32_t* __restrict__ buckets) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;  // < HALF by grid construction
  const int w = blockIdx.y;
  const uint32_t* cp = col_ptr + (size_t)w * (HALF + 1);
  const uint32_t begin = cp[slot], end = cp[slot + 1];
  const uint32_t* vi = val_idxs + (size_t)w * n;
  g1_xyzz acc = g1_identity();
  for (uint32_t t = begin; t < end; t++) {
    const uint32_t v = vi[t];
    const uint32_t* pt = bases + (size_t)(v & 0x7fffffffu) * 16;
    const fq px = ld_fq(pt);
    fq py = ld_fq(pt + 8);
    if (v >> 31) py = fq_neg_canonical(py);
    g1_madd(acc, px, py);
  }
  st_jacobian(buckets + ((size_t)w * HALF + slot) * 24, acc);
}

// ------------------------------------------------------------------------------------------------ stage 4: bucket reduce
// S_w = sum_{k=1}^{h-1} k * B[k] + h * B[0]   (≙ bpr.template.wgsl:38-132, CPU models test/utils.rs:222-338).
// Slot 0 carries weight h = 2^15, so it is treated as position 2^15: run j of length BPR_RUN covers positions
// j*RUN+1 .. (j+1)*RUN (position q reads slot q & 32767).  Per run: descending running sum (m, g), then
// g += (j*RUN) * m by double-and-add (stage_2 of the reference), then a workgroup tree reduction in LDS.
constexpr int BPR_RUN = 16;
constexpr int BPR_THREADS = HALF / BPR_RUN;  // 2048 runs per window
constexpr int BPR_BLOCK = 256;
constexpr int BPR_BLOCKS = BPR_THREADS / BPR_BLOCK;  // 8 partial sums per window

__global__ void __launch_bounds__(BPR_BLOCK) k_bpr_runs(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ partials) {
  __shared__ uint32_t lds[BPR_BLOCK * XYZZ_WORDS];
  const int w = blockIdx.y;
  const int t = threadIdx.x;
  const int j = blockIdx.x * BPR_BLOCK + t;  // run index, 0 .. 2047
  const uint32_t* bw = buckets + (size_t)w * HALF * 24;
  g1_xyzz m = g1_identity(), g = g1_identity();
  for (int q = (j + 1) * BPR_RUN; q > j * BPR_RUN; q--) {
    const g1_xyzz b = ld_jacobian(bw + (size_t)(q & (HALF - 1)) * 24);
    m = g1_add(m, b);
    g = g1_add(g, m);
  }
  const uint32_t s = (uint32_t)j * BPR_RUN;  // < 2^15
  if (s != 0 && !m.inf) {
    g1_xyzz sm = g1_identity();
    for (int bit = 14; bit >= 0; bit--) {
      sm = g1_double(sm);
      if ((s >> bit) & 1u) sm = g1_add(sm, m);
    }
    g = g1_add(g, sm);
  }
  st_xyzz(lds + t * XYZZ_WORDS, g);
  __syncthreads();
  for (int stride = BPR_BLOCK / 2; stride >= 1; stride >>= 1) {
    if (t < stride) {
      const g1_xyzz a = ld_xyzz(lds + t * XYZZ_WORDS);
      const g1_xyzz b = ld_xyzz(lds + (t + stride) * XYZZ_WORDS);
      st_xyzz(lds + t * XYZZ_WORDS, g1_add(a, b));
    }
    __syncthreads();
  }
  if (t == 0) {
    uint32_t* out = partials + ((size_t)w * BPR_BLOCKS + blockIdx.x) * XYZZ_WORDS;
    for (int i = 0; i < XYZZ_WORDS; i++) out[i] = lds[i];
  }
}

// one lane per window: add the BPR_BLOCKS partial sums, emit the window sum as canonical Jacobian bytes
__global__ void __launch_bounds__(64) k_bpr_final(const uint32_t* __restrict__ partials, int w_count, uint32_t* __restrict__ wsums) {
  const int w = threadIdx.x;
  if (w >= w_count) return;
  g1_xyzz acc = g1_identity();
  for (int b = 0; b < BPR_BLOCKS; b++) acc = g1_add(acc, ld_xyzz(partials + ((size_t)w * BPR_BLOCKS + b) * XYZZ_WORDS));
  st_jacobian_plain(wsums + (size_t)w * 24, acc);
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
__device__ __forceinline__ void draw256(uint64_t seed, uint64_t index, uint64_t attempt, uint64_t domain, uint32_t w[8]) {
  uint64_t base = splitmix64(seed ^ ((domain & 0xFF) << 56)) ^ (index * 0xD1342543DE82EF95ull);
  base = splitmix64(base ^ (attempt * 0xA0761D6478BD642Full));
  uint64_t s = base;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s = splitmix64(s);
    w[2 * i] = (uint32_t)s;
    w[2 * i + 1] = (uint32_t)(s >> 32);
  }
  w[7] &= 0x3FFFFFFFu;  // 254 bits
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

__device__ __forceinline__ fq fq_sqrt_candidate(const fq& a) {  // a^((p+1)/4), a exact
  fq acc = fq_one();
  for (int bit = 253; bit >= 0; bit--) {
    acc = fq_sqr(acc);
    if ((c_pp1d4[bit >> 5] >> (bit & 31)) & 1u) acc = fq_mul(acc, a);
  }
  return acc;
}

__global__ void __launch_bounds__(256) k_sample_points(uint64_t seed, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[8];
  for (uint64_t attempt = 0;; attempt++) {
    draw256(seed, i, attempt, 2, wx);
    if (geq_modulus<0>(wx)) continue;
    const fq x = fq_to_mont(fq_unpack(wx));
    const fq rhs = fq_canonical(fq_tidy(fq_add(fq_mul(fq_sqr(x), x), fq_three())));
    const fq y = fq_sqrt_candidate(rhs);
    if (!fq_equal_exact(fq_canonical(fq_sqr(y)), rhs)) continue;
    fq yp = fq_from_mont(y);  // canonical integer
    if ((yp.v[0] & 1u) != ((wx[0] >> 1) & 1u)) yp = fq_neg_canonical(yp);
    st8(out + i * 16, wx);
    st_fq(out + i * 16 + 8, yp);
    break;
  }
}

// ------------------------------------------------------------------------------------------------ op hooks for tests
// (≙ src/cuzk/wgsl/test/test_field.wgsl:13-62, test_point.wgsl:18-88)
__global__ void __launch_bounds__(256) k_test_fq(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fq x = fq_to_mont(ld_fq(a + i * 8));
  const fq y = b ? fq_to_mont(ld_fq(b + i * 8)) : fq_zero();
  fq z;
  switch (op) {
    case 0: z = fq_add(x, y); break;
    case 1: z = fq_sub<2>(x, y); break;
    case 2: z = fq_mul(x, y); break;
    case 3: z = fq_sqr(x); break;
    default: z = fq_neg_canonical(x); break;
  }
  st_fq(out + i * 8, fq_from_mont(z));
}

__global__ void __launch_bounds__(256) k_test_g1(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_xyzz p = ld_jacobian_plain(a + i * 24);
  g1_xyzz r;
  if (op == 0) {
    r = g1_add(p, ld_jacobian_plain(b + i * 24));
  } else if (op == 1) {
    r = g1_double(p);
  } else {
    const fq qx = fq_to_mont(ld_fq(b + i * 16)), qy = fq_to_mont(ld_fq(b + i * 16 + 8));
    g1_madd(p, qx, qy);
    r = p;
  }
  st_jacobian_plain(out + i * 24, r);
}

__global__ void __launch_bounds__(256) k_test_g1_mul_u32(const uint32_t* __restrict__ a, const uint32_t* __restrict__ k,
                                                         uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_jacobian_plain(out + i * 24, g1_mul_u32(ld_jacobian_plain(a + i * 24), k[i]));
}

}  // namespace msmk
