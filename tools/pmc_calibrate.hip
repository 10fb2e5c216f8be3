// Calibration kernels for reading rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 in THIS engine's access patterns
// (MI355X_MICROARCH.md, section HBM: "calibrate on a known byte count in your own access pattern").
//   k_stream_read   : 16 B per lane coalesced stream over `bytes`                    (known: bytes read)
//   k_gather64      : every lane reads one random 64-byte record as 4 x 16 B loads    (the SMVP point gather)
//   k_stream_write  : 16 B per lane coalesced stores                                  (known: bytes written)
//   k_scatter160    : every lane writes one 160-byte record as 10 x 16 B stores       (the SMVP bucket flush)
// Build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calibrate.hip -o /tmp/pmc_calibrate ; run under rocprofv3 --pmc ...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_stream_read(const uint4* __restrict__ in, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__global__ void k_gather64(const uint4* __restrict__ table, size_t records, size_t gathers, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < gathers; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = mix(i) % records;
    const uint4* p = table + r * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream_write(uint4* __restrict__ outp, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    outp[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void k_scatter160(uint4* __restrict__ outp, size_t records, size_t writes) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < writes; i += (size_t)gridDim.x * blockDim.x) {
    uint4* p = outp + (mix(i) % records) * 10;
#pragma unroll
    for (int k = 0; k < 10; k++) p[k] = make_uint4((uint32_t)i, k, 2, 3);
  }
}
int main() {
  const size_t table_bytes = (size_t)4 << 30;   // 4 GiB: far beyond the 256 MiB Infinity Cache
  const size_t moved = (size_t)1 << 30;         // every kernel moves 1 GiB
  uint4* buf; uint32_t* out;
  CK(hipMalloc(&buf, table_bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, table_bytes));
  for (int rep = 0; rep < 2; rep++) {
    k_stream_read<<<4096, 256>>>(buf, moved / 16, out);
    k_gather64<<<4096, 256>>>(buf, table_bytes / 64, moved / 64, out);
    k_stream_write<<<4096, 256>>>(buf, moved / 16);
    k_scatter160<<<4096, 256>>>(buf, table_bytes / 160, moved / 160);
  }
  CK(hipDeviceSynchronize());
  printf("calibration kernels done: each moved %zu bytes\n", moved);
  return 0;
}
