// Probe (gfx950): lane maps of v_mfma_i32_32x32x32_i8's A / B operands and the semantics of v_permlane32_swap, found with exact integer data.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_i8.hip -o gpurun_out/probe_mfma_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// A = one-hot at (lane la, byte ja) with value 1; B = all ones: C[row][*] = 1 for the row A's element belongs to -> its row; and the k is found
// with B one-hot at (lane lb, byte jb): C[row][col] = 1 iff k matches
__global__ void k_probe(int la, int ja, int lb, int jb, int* out) {
  const int l = threadIdx.x;
  unsigned char ab[16] = {}, bb[16] = {};
  if (l == la) ab[ja] = 1;
  if (lb < 0) { for (int j = 0; j < 16; j++) bb[j] = 1; } else if (l == lb) bb[jb] = 1;
  v4i a, b;
  memcpy(&a, ab, 16);
  memcpy(&b, bb, 16);
  v16i c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) out[l * 16 + r] = c[r];
}
__global__ void k_swap(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned x = 1000 + l, y = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[l * 2] = r[0];
  out[l * 2 + 1] = r[1];
}
int main() {
  int* d;
  hipMalloc(&d, 64 * 16 * 4);
  int h[64 * 16];
  // C layout assumed: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  auto rowcol = [&](int& row, int& col, int& cnt) {
    cnt = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++)
        if (h[l * 16 + r]) { cnt++; row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); col = l & 31; }
  };
  printf("A operand: (lane, byte) -> row (B all ones; count = 32 expected)\n");
  for (int la : {0, 1, 31, 32, 33, 63})
    for (int ja : {0, 1, 7, 8, 15}) {
      k_probe<<<1, 64>>>(la, ja, -1, 0, d);
      hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
      int row = -1, col = -1, cnt;
      rowcol(row, col, cnt);
      printf("  A lane %2d byte %2d -> row %2d (nonzero outputs %d)\n", la, ja, row, cnt);
    }
  printf("k index: A (lane 0, byte ja) meets B (lane lb, byte jb) where C[0][lb & 31] != 0\n");
  for (int ja = 0; ja < 16; ja++)
    for (int half = 0; half < 2; half++) {
      int found_lane = -1, found_byte = -1;
      for (int lbh = 0; lbh < 2 && found_lane < 0; lbh++)
        for (int jb = 0; jb < 16 && found_lane < 0; jb++) {
          k_probe<<<1, 64>>>(32 * half, ja, 32 * lbh + 5, jb, d);
          hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
          int row, col, cnt;
          rowcol(row, col, cnt);
          if (cnt) { found_lane = 32 * lbh + 5; found_byte = jb; }
        }
      printf("  A (lane %2d, byte %2d) <-> B (lane half %d, byte %2d)\n", 32 * half, ja, found_lane >> 5, found_byte);
    }
  unsigned* ds;
  hipMalloc(&ds, 64 * 2 * 4);
  unsigned hs[128];
  k_swap<<<1, 64>>>(ds);
  hipMemcpy(hs, ds, sizeof hs, hipMemcpyDeviceToHost);
  printf("permlane32_swap(x = 1000 + lane, y = 2000 + lane): lane 0 -> (%u, %u), lane 31 -> (%u, %u), lane 32 -> (%u, %u), lane 63 -> (%u, %u)\n", hs[0], hs[1], hs[62], hs[63],
         hs[64], hs[65], hs[126], hs[127]);
  return 0;
}
