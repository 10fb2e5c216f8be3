// Micro-benchmark (round 4, VERDICT r03 item 3 "C4"): the ceiling of the SMVP's memory access pattern on MI355X, without its arithmetic.
//
// k_smvp_chunks gathers one 64-byte packed affine point per (entry, lane) from the resident base array through an index list it reads
// in order -- 2^25 random 64-byte reads from 2 GiB per window at 2^24 points (endomorphism bases).  This program does exactly that and
// nothing else (usage: ubench_gather [multiply-adds per gather] [LDS pad bytes]): every lane walks a chunk of consecutive 4-byte indices and loads the 64 bytes they name (four 16-byte loads, one entry
// ahead, like the kernel), folds them into one word, and stores the word at the end.  Same launch shape as the SMVP: 256-thread
// workgroups, 168 VGPRs' worth of occupancy is imitated by the -DWAVES launch bound (3 waves per SIMD by default).
// Output: for table sizes from 64 MiB (Infinity Cache) to 2 GiB, the gather rate in G lines/s, ps per gather and GB/s.
//
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o /tmp/ubench_gather && /tmp/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef WAVES
#define WAVES 3
#endif

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

// pad: VGPRs held live across the loop so that the kernel's occupancy is the SMVP's (168 VGPRs -> 3 waves per SIMD)
__global__ void __launch_bounds__(256, WAVES) k_gather(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t chunk_len,
                                                      uint32_t chunks, uint32_t* __restrict__ out, int alu_per_entry) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= chunks) return;
  const uint32_t begin = c * chunk_len, end = begin + chunk_len, last = end - 1;
  uint32_t vnext = idx[begin], vnn = idx[begin + 1 < end ? begin + 1 : last];
  uint4 w0 = table[(size_t)vnext * 4], w1 = table[(size_t)vnext * 4 + 1], w2 = table[(size_t)vnext * 4 + 2], w3 = table[(size_t)vnext * 4 + 3];
  uint32_t acc = 0;
  for (uint32_t t = begin; t < end; t++) {
    const uint32_t x = w0.x ^ w0.y ^ w0.z ^ w0.w ^ w1.x ^ w1.y ^ w1.z ^ w1.w ^ w2.x ^ w2.y ^ w2.z ^ w2.w ^ w3.x ^ w3.y ^ w3.z ^ w3.w;
    vnext = vnn;
    vnn = idx[t + 2 < end ? t + 2 : last];
    const uint4* p = table + (size_t)vnext * 4;
    w0 = p[0];
    w1 = p[1];
    w2 = p[2];
    w3 = p[3];
    // optional stand-in for the arithmetic between two gathers: a dependent multiply-add chain of `alu_per_entry` steps
    uint64_t m = x;
    for (int k = 0; k < alu_per_entry; k++) m = m * 0x9E3779B97F4A7C15ull + acc;
    acc += (uint32_t)(m >> 7) ^ x;
  }
  out[c] = acc;
}

static uint64_t sm64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  const uint32_t chunk_len = 456;           // the SMVP's chunk length at 2^24 points x 8 windows
  const uint32_t chunks = 9u << 16;         // ~ SMVP_TARGET_LANES lanes
  const size_t entries = (size_t)chunk_len * chunks;  // 2.69e8: one launch = one 2^24 MSM's worth of gathers
  const int alu = argc > 1 ? atoi(argv[1]) : 0;
  // unused dynamic LDS per workgroup sets the occupancy: 160 KiB per CU / 52 KiB = 3 workgroups = 3 waves per SIMD (the SMVP's); 0: up to 8
  const unsigned lds_pad = argc > 2 ? (unsigned)atoi(argv[2]) : 52 * 1024;
  std::vector<uint32_t> h_idx(entries);
  uint32_t *d_idx, *d_out;
  uint4* d_table;
  const size_t max_lines = (size_t)1 << 25;  // 2 GiB of 64-byte records
  CHECK(hipMalloc(&d_table, max_lines * 64));
  CHECK(hipMemset(d_table, 0x5a, max_lines * 64));
  CHECK(hipMalloc(&d_idx, entries * 4));
  CHECK(hipMalloc(&d_out, (size_t)chunks * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("# %zu gathers of 64 B per launch (chunk %u x %u lanes), %u B of LDS pad per workgroup (%u workgroups per CU), %d dependent 64-bit multiply-adds per gather\n",
         entries, chunk_len, chunks, lds_pad, lds_pad ? 160 * 1024 / lds_pad : 8, alu);
  printf("# table_MiB  ms  G_gathers_per_s  ps_per_gather  GB_per_s\n");
  for (int lg = 20; lg <= 25; lg++) {  // 2^20 .. 2^25 records = 64 MiB .. 2 GiB
    const size_t lines = (size_t)1 << lg;
    uint64_t s = 12345 + lg;
    for (size_t i = 0; i < entries; i++) h_idx[i] = (uint32_t)(sm64(s) & (lines - 1));
    CHECK(hipMemcpy(d_idx, h_idx.data(), entries * 4, hipMemcpyHostToDevice));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_gather, dim3((chunks + 255) / 256), dim3(256), lds_pad, 0, d_table, d_idx, chunk_len, chunks, d_out, alu);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("%6zu  %8.3f  %7.2f  %6.1f  %7.1f\n", lines * 64 >> 20, best, entries / best / 1e6, best * 1e9 / entries, entries * 64.0 / best / 1e6);
    fflush(stdout);
  }
  return 0;
}
