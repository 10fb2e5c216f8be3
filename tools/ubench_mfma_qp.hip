// Micro-benchmark: a BN254 Fq Montgomery multiplication whose q * p product runs on the MATRIX cores --
// v_mfma_i32_32x32x32_i8 over a wave's 64 elements, the limb <-> digit re-layout included -- against the product's multiplier
// (int29_asm: 9 x 29-bit limbs, 162 v_mad_u64_u32, csrc/fq29_asm.h) in the same harness and with the same exactness check.
//
// Scheme "mfma_qp" (R = 2^264 = 33 bytes, so that the reduction is byte-aligned):
//   1. T = a * b            81 v_mad_u64_u32 on 9 x 29-bit limbs, carried into 18 limbs, packed into 17 32-bit words
//   2. q = T * N' mod R     N' = -p^-1 mod 2^264: low half product (9 x 29-bit limbs of T mod 2^261 ... 2^264: 10 limbs -> 55 multiply-adds)
//   3. q -> 34 SIGNED 8-bit digits: q + 0x80..80 as one multiword addition, every byte XOR 0x80 (the bytes are the B operand as they are)
//   4. U = q * p            as a matrix product on the matrix cores: C[k][e] = sum_i Toeplitz(p)[k][i] * qdigit_e[i]  (k: output byte column,
//                           e: the wave's element): A = the constant Toeplitz matrix of p's signed digits (2 M-tiles x 2 k-steps of 32: only byte columns 29 .. 66 are read),
//                           B = the wave's digits (2 N-tiles of 32 elements x 2 k-steps): 8 v_mfma_i32_32x32x32_i8 per wave-multiplication.
//                           Lane maps found with exact data (tools/probe_mfma_i8.hip): A / B byte j of lane half h <-> the same k; C: col = lane & 31,
//                           row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Both re-layouts are v_permlane32_swap only (no LDS): one swap per
//                           register pair gives every lane its own element's operand halves / output rows.
//   5. r = (T + U) / R      the high byte columns of U (k >= 33) evaluated with carries into 32-bit words + T's high words + m, where
//                           m = (T_lo + U_lo) / R is an exact integer recovered from U's top four low columns and T_lo's top bits; unpacked
//                           into 29-bit limbs for the next multiplication.
// Reported: ns per wave-multiplication per SIMD at 1 .. 4 waves per SIMD (one dependent chain per lane: x <- x * b), after an exact check of
// every lane against host big-integer arithmetic.   Build + run: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_mfma_qp.hip -o /tmp/ubench_mfma_qp && /tmp/ubench_mfma_qp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../msm-webgpu_amd/csrc/fq29.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using bn254::fq;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int L = 9, W = 29;
constexpr uint32_t MASK = (1u << W) - 1;

// constants of the scheme, filled by the host
__device__ uint32_t d_np29[10];      // N' = -p^-1 mod 2^264 as 10 limbs of 29 bits (the top one 3 bits)
constexpr int COL0 = 29;  // the first byte column of U that the reduction reads (columns 0 .. 28 cancel against T by construction; 29 .. 32 give m)
__device__ uint32_t d_afrag[2][2][64][4];  // A operand: [M-tile][k-step][lane][4 words = 16 signed digits of Toeplitz(p)]: rows = columns 29 .. 92

// ------------------------------------------------------------------------------------------------ the scheme
struct w17 { uint32_t w[17]; };

__device__ __forceinline__ fq mul_mfma(const fq& a, const fq& b, const v4i (&afr)[2][2]) {
  // 1. T = a * b: 18 limbs of 29 bits
  uint64_t c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < L; i++)
#pragma unroll
    for (int j = 0; j < L; j++) c[i + j] += (uint64_t)a.v[j] * b.v[i];
  uint32_t t[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) {
    t[k] = (uint32_t)c[k] & MASK;
    if (k + 1 < 2 * L) c[k + 1] += c[k] >> W;
  }
  // (operands < 2p < 2^255: T < 2^510, limb 17 < 2^17)
  // 2. q = T * N' mod 2^264: 10 limbs (29 x 9 = 261 bits + 3)
  uint64_t cq[10];
#pragma unroll
  for (int k = 0; k < 10; k++) cq[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++)
#pragma unroll
    for (int j = 0; i + j < 10; j++) cq[i + j] += (uint64_t)t[j] * d_np29[i];
  uint32_t q[10];
#pragma unroll
  for (int k = 0; k < 10; k++) {
    q[k] = (uint32_t)cq[k] & MASK;
    if (k + 1 < 10) cq[k + 1] += cq[k] >> W;
  }
  q[9] &= 7u;  // bits 261 .. 263
  // 3. q as 9 32-bit words (bits 0 .. 287, the top 24 zero), + 0x80 in every one of the 33 digit bytes, bytes XOR 0x80: signed digits
  uint32_t qw[9];
  {
    uint64_t acc = 0;
    int have = 0, k = 0;
#pragma unroll
    for (int wi = 0; wi < 9; wi++) {
      while (have < 32 && k < 10) {
        acc |= (uint64_t)q[k] << have;
        have += k < 9 ? W : 3;
        k++;
      }
      qw[wi] = (uint32_t)acc;
      acc >>= 32;
      have -= 32;
      if (have < 0) have = 0;
    }
  }
  {
    uint64_t carry = 0;
#pragma unroll
    for (int wi = 0; wi < 9; wi++) {
      const uint32_t bias = wi < 8 ? 0x80808080u : 0x00000080u;  // bytes 0 .. 32
      carry += (uint64_t)qw[wi] + bias;
      qw[wi] = (uint32_t)carry ^ (wi < 8 ? 0x80808080u : 0x00000080u);
      carry >>= 32;
    }
    // byte 33 (second byte of word 8) receives the final carry of the biased sum: it is already in place (sum of bytes 32 and carry-out:
    // q < 2^264 means byte 33 = carry out of byte 32, a digit 0 or 1)
  }
  // digits d = 0 .. 63 (34 .. 63 zero): k-step ks holds digits 32 ks .. 32 ks + 31; lane half h uses bytes 16 h .. 16 h + 15 of them
  v4i f0[2], f1[2];  // this element's bytes for half 0 / half 1 of each k-step
  f0[0] = v4i{(int)qw[0], (int)qw[1], (int)qw[2], (int)qw[3]};
  f1[0] = v4i{(int)qw[4], (int)qw[5], (int)qw[6], (int)qw[7]};
  f0[1] = v4i{(int)qw[8], 0, 0, 0};
  f1[1] = v4i{0, 0, 0, 0};
  // every lane's B operands: N-tile 0 (elements 0 .. 31) and 1 (32 .. 63) -- one permlane32 swap per register pair
  v4i bop[2][2];  // [N-tile][k-step]
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const auto s = __builtin_amdgcn_permlane32_swap((unsigned)f0[ks][r], (unsigned)f1[ks][r], false, false);
      bop[0][ks][r] = (int)s[0];
      bop[1][ks][r] = (int)s[1];
    }
  // 4. the matrix product and the swap that brings every element's 96 byte columns to its own lane
  int col[64];  // col[j] = byte column COL0 + j of U
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    v16i c0 = {}, c1 = {};
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[mt][0], bop[0][0], c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[mt][1], bop[0][1], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[mt][0], bop[1][0], c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[mt][1], bop[1][1], c1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const auto s = __builtin_amdgcn_permlane32_swap((unsigned)c0[r], (unsigned)c1[r], false, false);
      col[32 * mt + (r & 3) + 8 * (r >> 2)] = (int)s[0];      // rows 0-3, 8-11, ... of this element
      col[32 * mt + 4 + (r & 3) + 8 * (r >> 2)] = (int)s[1];  // rows 4-7, 12-15, ...
    }
  }
  // 5. r = (T + U) / 2^264.  T as bytes: T_lo = T mod 2^264, T_hi = T >> 264 (< 2^246).  m = (T_lo + U_lo) / 2^264 exactly, from the top
  //    columns: U_lo / 2^264 = col[32] / 2^8 + col[31] / 2^16 + col[30] / 2^24 + col[29] / 2^32 + (|rest| < 2^-12), T_lo / 2^264 from its top 32 bits
  //    T in 32-bit words (17 words = 544 bits)
  uint32_t tw[17];
  {
    uint64_t acc = 0;
    int have = 0, k = 0;
#pragma unroll
    for (int wi = 0; wi < 17; wi++) {
      while (have < 32 && k < 2 * L) {
        acc |= (uint64_t)t[k] << have;
        have += W;
        k++;
      }
      tw[wi] = (uint32_t)acc;
      acc >>= 32;
      have -= 32;
      if (have < 0) have = 0;
    }
  }
  // top 32 bits of T_lo = bits 232 .. 263 of T: word 7 (bits 224 .. 255) and word 8 (256 .. 287)
  const uint32_t tlo_top = (tw[7] >> 8) | (tw[8] << 24);
  // fixed point with 32 fraction bits: m = round(tlo_top + col[32] 2^24 + col[31] 2^16 + col[30] 2^8 + col[29]) / 2^32
  const int64_t frac = (int64_t)tlo_top + ((int64_t)col[32 - COL0] << 24) + ((int64_t)col[31 - COL0] << 16) + ((int64_t)col[30 - COL0] << 8) + (int64_t)col[29 - COL0];
  const int64_t m = (frac + ((int64_t)1 << 31)) >> 32;
  // high part: sum_{k >= 33} col[k] 256^(k - 33) + T_hi + m, as 32-bit words with signed carries
  uint32_t rw[9];
  {
    int64_t acc = m;
#pragma unroll
    for (int wi = 0; wi < 9; wi++) {
      // T_hi word wi = bits 264 + 32 wi .. of T = (tw[8 + wi] >> 8) | (tw[9 + wi] << 24)
      const uint32_t thi = (8 + wi < 17 ? tw[8 + wi] >> 8 : 0u) | (9 + wi < 17 ? tw[9 + wi] << 24 : 0u);
      acc += (int64_t)thi;
#pragma unroll
      for (int bb = 0; bb < 4; bb++) {
        const int k = 33 + 4 * wi + bb;
        if (k < 67) acc += (int64_t)col[k - COL0] << (8 * bb);  // 34 + 34 signed digits: byte columns 0 .. 66, the rest are zero
      }
      rw[wi] = (uint32_t)acc;
      acc >>= 32;  // arithmetic: signed carry
    }
  }
  // unpack into 9 limbs of 29 bits (the result is < 2p: 255 bits)
  fq r;
  {
    uint64_t acc = 0;
    int have = 0, wi = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
      while (have < W && wi < 9) {
        acc |= (uint64_t)rw[wi] << have;
        have += 32;
        wi++;
      }
      r.v[k] = (uint32_t)acc & MASK;
      acc >>= W;
      have -= W;
    }
  }
  return r;
}

template <int SCHEME>  // 0: int29_asm (the product's multiplier, R = 2^261), 1: mfma_qp (R = 2^264)
__global__ void __launch_bounds__(256) k_mul(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) % n;
  fq a, b;
#pragma unroll
  for (int k = 0; k < L; k++) {
    a.v[k] = in[(2 * i) * L + k];
    b.v[k] = in[(2 * i + 1) * L + k];
  }
  if constexpr (SCHEME == 0) {
#if defined(FQ29_ASM)
    for (int it = 0; it < iters; it++) a = bn254::fq_mul_asm(a, b);  // device pass only
#endif
  } else {
    v4i afr[2][2];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) afr[mt][ks] = v4i{(int)d_afrag[mt][ks][lane][0], (int)d_afrag[mt][ks][lane][1], (int)d_afrag[mt][ks][lane][2], (int)d_afrag[mt][ks][lane][3]};
    for (int it = 0; it < iters; it++) a = mul_mfma(a, b, afr);
  }
  if ((size_t)blockIdx.x * blockDim.x + threadIdx.x < n)
#pragma unroll
    for (int k = 0; k < L; k++) out[i * L + k] = a.v[k];
}

// ------------------------------------------------------------------------------------------------ host big integers (little-endian 32-bit words)
typedef std::vector<uint32_t> big;
static big mul(const big& a, const big& b) {
  big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t carry = 0;
    for (size_t j = 0; j < b.size(); j++) {
      const uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    r[i + b.size()] = (uint32_t)carry;
  }
  return r;
}
static big add(const big& a, const big& b) {
  big r(std::max(a.size(), b.size()) + 1, 0);
  uint64_t c = 0;
  for (size_t i = 0; i < r.size(); i++) {
    c += (uint64_t)(i < a.size() ? a[i] : 0) + (i < b.size() ? b[i] : 0);
    r[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}
static big low_bits(big a, int bits) {
  a.resize((bits + 31) / 32, 0);
  if (bits % 32) a.back() &= (1u << (bits % 32)) - 1;
  return a;
}
static big shr(const big& a, int bits) {
  big r;
  for (size_t i = bits / 32; i < a.size(); i++) {
    uint64_t v = a[i] >> (bits % 32);
    if (bits % 32 && i + 1 < a.size()) v |= (uint64_t)a[i + 1] << (32 - bits % 32);
    r.push_back((uint32_t)v);
  }
  return r;
}
static big from_limbs29(const uint32_t* v, int n) {
  big r((29 * n + 31) / 32 + 1, 0);
  for (int k = 0; k < n; k++) {
    const int bit = 29 * k;
    const uint64_t x = (uint64_t)v[k] << (bit % 32);
    r[bit / 32] += (uint32_t)x;  // limbs are < 2^29 and do not overlap: no carries
    r[bit / 32 + 1] += (uint32_t)(x >> 32);
  }
  return r;
}
static bool equal(big a, big b) {
  while (!a.empty() && a.back() == 0) a.pop_back();
  while (!b.empty() && b.back() == 0) b.pop_back();
  return a == b;
}

int main() {
  // p (bn254 Fq) as words, N' = -p^-1 mod 2^264 by Newton iteration on 2^k
  big p(8);
  for (int i = 0; i < 8; i++) p[i] = bn254::FQ_P32[i];
  // inverse of p mod 2^264: x <- x (2 - p x), doubling the precision
  big x = {1};  // p is odd: p^-1 = 1 mod 2
  for (int bits = 1; bits < 264; bits *= 2) {
    const int nb = std::min(2 * bits, 264);
    big px = low_bits(mul(p, x), nb);
    // two = 2 - px mod 2^nb
    big two(px.size(), 0);
    uint64_t borrow = 0;
    for (size_t i = 0; i < px.size(); i++) {
      const uint64_t s = (uint64_t)(i == 0 ? 2 : 0) - px[i] - borrow;
      two[i] = (uint32_t)s;
      borrow = (s >> 32) & 1;
    }
    x = low_bits(mul(x, low_bits(two, nb)), nb);
  }
  // N' = 2^264 - x
  big np(9, 0);
  {
    uint64_t borrow = 0;
    for (size_t i = 0; i < 9; i++) {
      const uint64_t s = (uint64_t)0 - (i < x.size() ? x[i] : 0) - borrow;
      np[i] = (uint32_t)s;
      borrow = (s >> 32) & 1;
    }
    np = low_bits(np, 264);
  }
  if (!equal(low_bits(add(mul(p, np), big{1}), 264), big{})) { printf("N' is wrong\n"); return 1; }
  uint32_t np29[10];
  for (int k = 0; k < 10; k++) {
    const int bit = 29 * k;
    uint64_t v = np[bit / 32] >> (bit % 32);
    if (bit / 32 + 1 < 9) v |= (uint64_t)np[bit / 32 + 1] << (32 - bit % 32);
    np29[k] = (uint32_t)v & (k < 9 ? MASK : 7u);
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(d_np29), np29, sizeof np29));
  // p's signed 8-bit digits: bytes of p + 0x80..80 (33 bytes), each minus 128; digit 33 = the carry out
  int pd[34];
  {
    big pb = p;
    pb.resize(9, 0);
    uint64_t carry = 0;
    uint8_t bytes[36];
    for (int wi = 0; wi < 9; wi++) {
      const uint32_t bias = wi < 8 ? 0x80808080u : 0x00000080u;
      carry += (uint64_t)pb[wi] + bias;
      const uint32_t s = (uint32_t)carry;
      carry >>= 32;
      memcpy(bytes + 4 * wi, &s, 4);
    }
    for (int d = 0; d < 33; d++) pd[d] = (int)bytes[d] - 128;
    pd[33] = bytes[33];
    // check: sum pd[d] 256^d == p
  }
  // A fragments: A[row k][digit index i] = pd[k - i]; lane l holds row (l & 31) of its M-tile, bytes j = 0 .. 15 <-> i = 32 ks + 16 (l >> 5) + j
  static uint32_t afrag[2][2][64][4];
  for (int mt = 0; mt < 2; mt++)
    for (int ks = 0; ks < 2; ks++)
      for (int l = 0; l < 64; l++) {
        int8_t b16[16];
        for (int j = 0; j < 16; j++) {
          const int k = COL0 + 32 * mt + (l & 31), i = 32 * ks + 16 * (l >> 5) + j, d = k - i;
          b16[j] = (i < 34 && d >= 0 && d < 34) ? (int8_t)pd[d] : 0;
        }
        memcpy(afrag[mt][ks][l], b16, 16);
      }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(d_afrag), afrag, sizeof afrag));

  // inputs: pairs (a, b) below p as 29-bit limbs
  const size_t n = 16384;
  std::vector<uint32_t> in(2 * n * L);
  uint64_t s = 0x1234567887654321ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t i = 0; i < 2 * n; i++) {
    for (int k = 0; k < L; k++) in[i * L + k] = (uint32_t)rnd() & MASK;
    in[i * L + L - 1] &= (1u << (253 - 29 * 8)) - 1;  // < 2^253 < p
  }
  uint32_t *d_in, *d_out;
  CK(hipMalloc(&d_in, in.size() * 4));
  CK(hipMalloc(&d_out, n * L * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  // exactness: ONE multiplication per lane against the host: r R == a b + q p with q = (a b) N' mod R  <=>  r == (a b + q p) >> shift
  for (int scheme = 0; scheme < 2; scheme++) {
    const int shift = scheme == 0 ? 261 : 264;
    if (scheme == 0) hipLaunchKernelGGL(k_mul<0>, dim3(n / 256), dim3(256), 0, 0, d_in, d_out, 1, n);
    else hipLaunchKernelGGL(k_mul<1>, dim3(n / 256), dim3(256), 0, 0, d_in, d_out, 1, n);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> out(n * L);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    // N' for this R
    big npr = scheme == 1 ? np : big();
    if (scheme == 0) {  // -p^-1 mod 2^261 = N'(264) mod 2^261
      npr = low_bits(np, 261);
    }
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
      const big a = from_limbs29(&in[(2 * i) * L], L), b = from_limbs29(&in[(2 * i + 1) * L], L);
      const big t = mul(a, b);
      const big q = low_bits(mul(low_bits(t, shift), npr), shift);
      const big want = shr(add(t, mul(q, p)), shift);
      if (!equal(want, from_limbs29(&out[i * L], L))) bad++;
    }
    printf("%s: exactness over %zu products: %s (%zu wrong)\n", scheme == 0 ? "int29_asm (R = 2^261)" : "mfma_qp   (R = 2^264)", n, bad ? "FAILED" : "ok", bad);
    if (bad) return 1;
  }
  // timing: ns per wave-multiplication per SIMD at 1 .. 4 waves per SIMD (blocks of 256 threads = one wave per SIMD of a CU)
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("ns per wave-multiplication per SIMD (dependent chain, %d iterations):\n  waves/SIMD   int29_asm    mfma_qp    ratio\n", iters);
  for (int waves = 1; waves <= 4; waves++) {
    float ms[2];
    for (int scheme = 0; scheme < 2; scheme++) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        if (scheme == 0) hipLaunchKernelGGL(k_mul<0>, dim3(cus * waves), dim3(256), 0, 0, d_in, d_out, iters, n);
        else hipLaunchKernelGGL(k_mul<1>, dim3(cus * waves), dim3(256), 0, 0, d_in, d_out, iters, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (t < best) best = t;
      }
      ms[scheme] = best;
    }
    const double ns0 = ms[0] * 1e6 / ((double)iters * waves), ns1 = ms[1] * 1e6 / ((double)iters * waves);
    printf("  %d            %8.1f   %8.1f    %.2f x slower\n", waves, ns0, ns1, ns1 / ns0);
  }
  return 0;
}
