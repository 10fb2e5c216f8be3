// Micro-benchmark: BN254 Fq Montgomery multiplication on gfx950 (MI355X), three limb schemes in ONE harness:
//   int29_cxx   9 x 29-bit limbs, v_mad_u64_u32, the C++ form of csrc/fq29.h           (R = 2^261)
//   int29_asm   the same product as one inline-assembly block, csrc/fq29_asm.h           (R = 2^261)   <- the SMVP's multiplier
//   fp64_5x52   5 x 52-bit limbs held as doubles, every 52 x 52 partial product split into high and low half by two
//               v_fma_f64 in round-toward-zero mode (N. Emmart's scheme: hi = fma(a, b, 2^104); lo = fma(a, b, 2^104 + 2^52 - hi)),
//               column sums accumulated as 64-bit integers of the doubles' bit patterns           (R = 2^260)
// Reported: ns per wave-multiplication per SIMD at 1..4 waves per SIMD (dependent chains, two independent chains per lane),
// after an exact check of every scheme against host big-integer arithmetic (a * b == c * R mod p).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_fqmul.hip -o gpurun_out/ubench_fqmul
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../msm-webgpu_amd/csrc/fq29.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using bn254::fq;

__device__ __forceinline__ fq mul_asm(const fq& a, const fq& b) {
#if defined(FQ29_ASM)
  return bn254::fq_mul_asm(a, b);  // device pass only
#else
  return bn254::fq_mul(a, b);
#endif
}

// ------------------------------------------------------------------------------------------------ FP64 scheme
struct fq52 {
  double v[5];
};
constexpr uint64_t MASK52 = (1ull << 52) - 1;
constexpr uint64_t BITS_2P52 = 0x4330000000000000ull;   // bit pattern of 2^52  (+ m: the double 2^52 + m, m < 2^52)
constexpr uint64_t BITS_2P104 = 0x4670000000000000ull;  // bit pattern of 2^104 (+ h: the double 2^104 + h * 2^52)
// p in 52-bit limbs and n0 = -p^-1 mod 2^52, filled at start-up from the host (see main)
__device__ double d_p52[5];
__device__ double d_n0_52;

__device__ __forceinline__ double fma_rz(double a, double b, double c) {  // rounding mode comes from the MODE register
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double sub_f64(double a, double b) {
  double d;
  asm("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }
__device__ __forceinline__ double from_bits(uint64_t x) { return __longlong_as_double((long long)x); }

// a * b / 2^260 mod p; operands: integer-valued doubles in [0, 2^52); result likewise, value < 2p
__device__ __forceinline__ fq52 mont52(const fq52& a, const fq52& b, const double p52[5], double n0) {
  const double C1 = from_bits(BITS_2P104), C2 = from_bits(BITS_2P104 + 1), TWO52 = from_bits(BITS_2P52);  // 2^104, 2^104 + 2^52, 2^52
  // column k receives, with statically known counts, `nlo` low halves (bias 2^52 pattern) and `nhi` high halves (bias 2^104
  // pattern): start every column at minus its total bias, except the low half of q_k * p_0, which is taken off at the carry
  uint64_t col[11];
#pragma unroll
  for (int k = 0; k < 11; k++) {
    int nlo = 0, nhi = 0;
    for (int i = 0; i < 5; i++)
      for (int j = 0; j < 5; j++) {
        if (i + j == k) nlo += 2;      // a_j b_i and q_i p_j
        if (i + j + 1 == k) nhi += 2;
      }
    if (k < 5) nlo -= 1;
    col[k] = 0ull - ((uint64_t)nlo * BITS_2P52 + (uint64_t)nhi * BITS_2P104);
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double hi = fma_rz(a.v[j], b.v[i], C1);
      const double lo = fma_rz(a.v[j], b.v[i], sub_f64(C2, hi));
      col[i + j] += bits(lo);
      col[i + j + 1] += bits(hi);
    }
    // q = col[i] * n0 mod 2^52
    const double xd = sub_f64(from_bits((col[i] & MASK52) | BITS_2P52), TWO52);
    const double hq = fma_rz(xd, n0, C1);
    const double q = sub_f64(fma_rz(xd, n0, sub_f64(C2, hq)), TWO52);
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double hi = fma_rz(q, p52[j], C1);
      const double lo = fma_rz(q, p52[j], sub_f64(C2, hi));
      col[i + j] += bits(lo);
      col[i + j + 1] += bits(hi);
    }
    col[i + 1] += (col[i] - BITS_2P52) >> 52;  // col[i] is now a multiple of 2^52
  }
  fq52 r;
#pragma unroll
  for (int k = 5; k < 10; k++) {
    r.v[k - 5] = sub_f64(from_bits((col[k] & MASK52) | BITS_2P52), TWO52);
    col[k + 1] += col[k] >> 52;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------ kernels
enum Scheme { INT29_CXX, INT29_ASM, FP64_5X52 };
static const char* scheme_name[] = {"int29_cxx (9 x 29 bit, v_mad_u64_u32, C++)", "int29_asm (9 x 29 bit, v_mad_u64_u32, inline asm)",
                                    "fp64_5x52 (5 x 52 bit, v_fma_f64 hi/lo split)"};

__device__ __forceinline__ void set_round_toward_zero_f64() {
  // MODE register (hwreg 1), FP_ROUND bits [3:2] = f64/f16 rounding: 3 = toward zero
  __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);
}

// out[lane] = a[lane] (*) b[lane], `iters` times: x <- x * b (two independent chains per lane)
template <int SCHEME>
__global__ void __launch_bounds__(256) k_mul(const uint32_t* __restrict__ in29, const double* __restrict__ in52, uint32_t* __restrict__ out29,
                                             double* __restrict__ out52, int iters, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) % n;
  if constexpr (SCHEME == FP64_5X52) {
    set_round_toward_zero_f64();
    fq52 a, b, c;
    double p52[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      a.v[k] = in52[(2 * i) * 5 + k];
      b.v[k] = in52[(2 * i + 1) * 5 + k];
      p52[k] = d_p52[k];
    }
    const double n0 = d_n0_52;
    c = b;
    for (int it = 0; it < iters; it++) {
      a = mont52(a, b, p52, n0);
      if (iters > 1) c = mont52(c, b, p52, n0);
    }
    if ((size_t)blockIdx.x * blockDim.x + threadIdx.x < n) {
#pragma unroll
      for (int k = 0; k < 5; k++) out52[i * 5 + k] = iters > 1 ? a.v[k] + c.v[k] : a.v[k];  // timing runs: keep both chains alive
    }
  } else {
    fq a, b, c;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      a.v[k] = in29[(2 * i) * 9 + k];
      b.v[k] = in29[(2 * i + 1) * 9 + k];
    }
    c = b;
    for (int it = 0; it < iters; it++) {
      if constexpr (SCHEME == INT29_ASM) {
        a = mul_asm(a, b);
        if (iters > 1) c = mul_asm(c, b);
      } else {
        a = bn254::fq_mul(a, b);
        if (iters > 1) c = bn254::fq_mul(c, b);
      }
    }
    if ((size_t)blockIdx.x * blockDim.x + threadIdx.x < n) {
#pragma unroll
      for (int k = 0; k < 9; k++) out29[i * 9 + k] = iters > 1 ? (a.v[k] ^ c.v[k]) : a.v[k];  // timing runs: keep both chains alive
    }
  }
}

// ------------------------------------------------------------------------------------------------ host big integers (4 x 64 bit)
struct u256 {
  uint64_t w[4];
};
static const u256 P = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static bool geq(const u256& a, const u256& b) {
  for (int i = 3; i >= 0; i--)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
  return true;
}
static u256 sub(const u256& a, const u256& b) {
  u256 r;
  unsigned __int128 br = 0;
  for (int i = 0; i < 4; i++) {
    unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - br;
    r.w[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  return r;
}
static u256 addmod(const u256& a, const u256& b) {  // a, b < p
  u256 r;
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.w[i] + b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return geq(r, P) ? sub(r, P) : r;  // p < 2^254: no carry out
}
static u256 mulmod(const u256& a, const u256& b) {  // double-and-add
  u256 r = {{0, 0, 0, 0}};
  for (int bit = 255; bit >= 0; bit--) {
    r = addmod(r, r);
    if ((b.w[bit >> 6] >> (bit & 63)) & 1) r = addmod(r, a);
  }
  return r;
}
static u256 pow2mod(int e) {
  u256 r = {{1, 0, 0, 0}};
  for (int i = 0; i < e; i++) r = addmod(r, r);
  return r;
}
static u256 reduce(u256 x) {  // x < 2^256 -> x mod p (p > 2^253)
  while (geq(x, P)) x = sub(x, P);
  return x;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static u256 rnd_fq() {
  for (;;) {
    u256 x = {{rnd(), rnd(), rnd(), rnd() & 0x3fffffffffffffffull}};
    if (!geq(x, P)) return x;
  }
}
static void to29(const u256& x, uint32_t v[9]) {
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, w = bit >> 6, s = bit & 63;
    uint64_t t = x.w[w] >> s;
    if (s > 35 && w < 3) t |= x.w[w + 1] << (64 - s);
    v[k] = (uint32_t)(t & 0x1fffffffu);
  }
}
static u256 from29(const uint32_t v[9]) {
  u256 r = {{0, 0, 0, 0}};
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, w = bit >> 6, s = bit & 63;
    r.w[w] |= (uint64_t)v[k] << s;
    if (s > 35 && w < 3) r.w[w + 1] |= (uint64_t)v[k] >> (64 - s);
  }
  return r;
}
static void to52(const u256& x, double v[5]) {
  for (int k = 0; k < 5; k++) {
    const int bit = 52 * k, w = bit >> 6, s = bit & 63;
    uint64_t t = x.w[w] >> s;
    if (s > 12 && w < 3) t |= x.w[w + 1] << (64 - s);
    v[k] = (double)(t & MASK52);
  }
}
static u256 from52(const double v[5]) {
  u256 r = {{0, 0, 0, 0}};
  for (int k = 0; k < 5; k++) {
    const uint64_t limb = (uint64_t)v[k];
    const int bit = 52 * k, w = bit >> 6, s = bit & 63;
    if (w < 4) r.w[w] |= limb << s;
    if (s > 12 && w < 3) r.w[w + 1] |= limb >> (64 - s);
  }
  return r;
}

template <int SCHEME>
static int bench(const uint32_t* d_in29, const double* d_in52, uint32_t* d_out29, double* d_out52, size_t n, float* ns_at) {
  const int iters = 1500;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int idx = 0;
  for (int waves : {1, 2, 3, 4}) {
    const int blocks = 256 * waves;
    k_mul<SCHEME><<<blocks, 256>>>(d_in29, d_in52, d_out29, d_out52, iters, n);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      CK(hipEventRecord(e0));
      k_mul<SCHEME><<<blocks, 256>>>(d_in29, d_in52, d_out29, d_out52, iters, n);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double per = best * 1e6 / ((double)waves * iters * 2);
    printf("%-52s waves/SIMD=%d  %8.3f ms  %8.1f ns per wave-multiplication per SIMD  (%.1f G modmul/s per GPU)\n", scheme_name[SCHEME], waves,
           best, per, 1024.0 * 64.0 / per);
    ns_at[idx++] = (float)per;
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs=%d  arch=%s\n", prop.name, prop.multiProcessorCount, prop.gcnArchName);
  // constants of the FP64 scheme
  {
    double p52[5];
    to52(P, p52);
    // n0 = -p^-1 mod 2^52 by Newton iteration on 64-bit words
    const uint64_t p0 = P.w[0];
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;  // p0 * inv == 1 mod 2^64
    const double n0 = (double)((0 - inv) & MASK52);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(d_p52), p52, sizeof(p52)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(d_n0_52), &n0, sizeof(n0)));
  }
  const size_t n = 1 << 14;
  std::vector<u256> A(n), B(n);
  std::vector<uint32_t> in29(n * 18), out29(n * 9);
  std::vector<double> in52(n * 10), out52(n * 5);
  for (size_t i = 0; i < n; i++) {
    A[i] = rnd_fq();
    B[i] = rnd_fq();
    if (i == 0) A[i] = {{0, 0, 0, 0}};
    if (i == 1) A[i] = sub(P, {{1, 0, 0, 0}}), B[i] = A[i];
    if (i == 2) A[i] = {{1, 0, 0, 0}};
    to29(A[i], &in29[(2 * i) * 9]);
    to29(B[i], &in29[(2 * i + 1) * 9]);
    to52(A[i], &in52[(2 * i) * 5]);
    to52(B[i], &in52[(2 * i + 1) * 5]);
  }
  uint32_t *d_in29, *d_out29;
  double *d_in52, *d_out52;
  CK(hipMalloc(&d_in29, in29.size() * 4));
  CK(hipMalloc(&d_out29, out29.size() * 4));
  CK(hipMalloc(&d_in52, in52.size() * 8));
  CK(hipMalloc(&d_out52, out52.size() * 8));
  CK(hipMemcpy(d_in29, in29.data(), in29.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_in52, in52.data(), in52.size() * 8, hipMemcpyHostToDevice));

  // ---- exactness: c * R == a * b (mod p) for every scheme
  const u256 R261 = pow2mod(261), R260 = pow2mod(260);
  const int blocks = (int)(n / 256);
  int bad[3] = {0, 0, 0};
  for (int scheme = 0; scheme < 3; scheme++) {
    if (scheme == INT29_CXX) k_mul<INT29_CXX><<<blocks, 256>>>(d_in29, d_in52, d_out29, d_out52, 1, n);
    if (scheme == INT29_ASM) k_mul<INT29_ASM><<<blocks, 256>>>(d_in29, d_in52, d_out29, d_out52, 1, n);
    if (scheme == FP64_5X52) k_mul<FP64_5X52><<<blocks, 256>>>(d_in29, d_in52, d_out29, d_out52, 1, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out29.data(), d_out29, out29.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out52.data(), d_out52, out52.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
      const u256 want = mulmod(A[i], B[i]);
      u256 c = scheme == FP64_5X52 ? from52(&out52[i * 5]) : from29(&out29[i * 9]);
      bool limbs_ok = true;
      if (scheme == FP64_5X52)
        for (int k = 0; k < 5; k++) limbs_ok = limbs_ok && out52[i * 5 + k] >= 0 && out52[i * 5 + k] < 4503599627370496.0 && out52[i * 5 + k] == (double)(uint64_t)out52[i * 5 + k];
      const u256 got = mulmod(reduce(c), scheme == FP64_5X52 ? R260 : R261);
      if (!limbs_ok || memcmp(&got, &want, sizeof(u256)) != 0) bad[scheme]++;
    }
    printf("exactness %-52s %zu products, %d wrong\n", scheme_name[scheme], n, bad[scheme]);
  }
  if (bad[0] || bad[1] || bad[2]) {
    printf("EXACTNESS CHECK FAILED\n");
    return 2;
  }
  float ns[3][4];
  if (bench<INT29_CXX>(d_in29, d_in52, d_out29, d_out52, n, ns[0])) return 1;
  if (bench<INT29_ASM>(d_in29, d_in52, d_out29, d_out52, n, ns[1])) return 1;
  if (bench<FP64_5X52>(d_in29, d_in52, d_out29, d_out52, n, ns[2])) return 1;
  printf("\nsummary at 3 waves/SIMD (the SMVP's occupancy): int29_cxx %.1f ns, int29_asm %.1f ns, fp64_5x52 %.1f ns  ->  fp64 / int29_asm = %.2f\n",
         ns[0][2], ns[1][2], ns[2][2], ns[2][2] / ns[1][2]);
  return 0;
}
