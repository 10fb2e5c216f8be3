// Micro-benchmark: VALU instruction issue rates on gfx950 (MI355X).
// Decides the limb scheme of the BN254 Fq multiplier (DESIGN.md "Field multiplier").
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2048;
constexpr int NACC = 8;      // independent chains per lane
constexpr int UNROLL = 4;    // ops per chain per iteration

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64, ADD_F64, MUL_F64, FMA_F32,
          PK_FMA_F32, ADD_CO_U32, ADDC_CO_U32, ADD3_U32, LSHL_ADD_U32, ALIGNBIT, AND_OR, MAD_U32_U16, ADD_U32, XAD_U32,
          MAD_I32_I24, LSHLREV_B64, CVT_F64_U32, ADD_LSHL, PERM_B32, MAD_U64_U32_DEP, NOPS };

static const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_fma_f64",
  "v_add_f64", "v_mul_f64", "v_fma_f32", "v_pk_fma_f32", "v_add_co_u32", "v_addc_co_u32", "v_add3_u32", "v_lshl_add_u32", "v_alignbit_b32",
  "v_and_or_b32", "v_mad_u32_u16", "v_add_u32", "v_xad_u32", "v_mad_i32_i24", "v_lshlrev_b64", "v_cvt_f64_u32", "v_add_lshl_u32", "v_perm_b32", "v_mad_u64_u32(dep1)"};

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x * 2654435761u, b = (seed ^ 0x9e3779b9u) + threadIdx.x;
  uint64_t acc[NACC];
  double dacc[NACC];
  float facc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) { acc[i] = a * (i + 1) + b; dacc[i] = (double)(a & 0xffff) + i; facc[i] = (float)i + 1.0f; }
  double da = (double)(a & 0xffffff) * 1e-9, db = (double)(b & 0xffffff) * 1e-9;
  float fa = (float)da, fb = (float)db;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
        if constexpr (OP == MAD_U64_U32) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc"); }
        else if constexpr (OP == MAD_U64_U32_DEP) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc"); }
        else if constexpr (OP == MUL_LO_U32) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == MUL_HI_U32) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == MAD_U32_U24) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        else if constexpr (OP == MAD_I32_I24) { asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        else if constexpr (OP == MUL_HI_U32_U24) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == FMA_F64) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db)); }
        else if constexpr (OP == ADD_F64) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(dacc[i]) : "v"(da)); }
        else if constexpr (OP == MUL_F64) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dacc[i]) : "v"(da)); }
        else if constexpr (OP == FMA_F32) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(facc[i]) : "v"(fa), "v"(fb)); }
        else if constexpr (OP == PK_FMA_F32) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db)); }
        else if constexpr (OP == ADD_CO_U32) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo) : "v"(a) : "vcc"); acc[i] = lo; }
        else if constexpr (OP == ADDC_CO_U32) { asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo) : "v"(a) : "vcc"); acc[i] = lo; }
        else if constexpr (OP == ADD3_U32) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        else if constexpr (OP == LSHL_ADD_U32) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == ADD_LSHL) { asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == ALIGNBIT) { asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == AND_OR) { asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        else if constexpr (OP == MAD_U32_U16) { asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        else if constexpr (OP == ADD_U32) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        else if constexpr (OP == XAD_U32) { asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        else if constexpr (OP == LSHLREV_B64) { asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(acc[i])); }
        else if constexpr (OP == CVT_F64_U32) { asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(dacc[i]) : "v"(lo)); acc[i] = lo + 1; }
        else if constexpr (OP == PERM_B32) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        (void)hi;
      }
    }
  }
  uint64_t s = 0; double ds = 0; float fs = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) { s += acc[i]; ds += dacc[i]; fs += facc[i]; }
  if (s == 0x123456789abcdefull && ds == 1.2345 && fs == 3.0f) out[threadIdx.x] = 1;  // never true; keeps results live
}

template <int OP>
int run(uint32_t* d, int waves_per_simd, double ref_clock_ghz) {
  int ncu = 256;
  int blocks = ncu * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(d, 12345);  // warm
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(d, 12345 + r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double wave_instrs_per_simd = (double)waves_per_simd * ITER * UNROLL * (OP == MAD_U64_U32_DEP ? NACC : NACC);
  double ns_per = best * 1e6 / wave_instrs_per_simd;
  printf("%-22s waves/SIMD=%d  %8.3f ms  %7.3f ns/wave-instr/SIMD  = %6.2f cyc @%.1fGHz\n", names[OP], waves_per_simd, best, ns_per,
         ns_per * ref_clock_ghz, ref_clock_ghz);
  return 0;
}

template <int OP>
int run_all(uint32_t* d) {
  for (int w : {1, 2, 4}) if (run<OP>(d, w, 2.4)) return 1;
  return 0;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs=%d  clock=%d kHz  arch=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
  uint32_t* d; CK(hipMalloc(&d, 4096));
  int rc = 0;
  rc |= run_all<FMA_F32>(d); rc |= run_all<PK_FMA_F32>(d);
  rc |= run_all<MAD_U64_U32>(d); rc |= run_all<MAD_U64_U32_DEP>(d); rc |= run_all<MUL_LO_U32>(d); rc |= run_all<MUL_HI_U32>(d);
  rc |= run_all<MAD_U32_U24>(d); rc |= run_all<MAD_I32_I24>(d); rc |= run_all<MUL_HI_U32_U24>(d); rc |= run_all<MAD_U32_U16>(d);
  rc |= run_all<FMA_F64>(d); rc |= run_all<ADD_F64>(d); rc |= run_all<MUL_F64>(d); rc |= run_all<CVT_F64_U32>(d);
  rc |= run_all<ADD_U32>(d); rc |= run_all<ADD_CO_U32>(d); rc |= run_all<ADDC_CO_U32>(d); rc |= run_all<ADD3_U32>(d);
  rc |= run_all<LSHL_ADD_U32>(d); rc |= run_all<ADD_LSHL>(d); rc |= run_all<ALIGNBIT>(d); rc |= run_all<AND_OR>(d); rc |= run_all<XAD_U32>(d);
  rc |= run_all<LSHLREV_B64>(d); rc |= run_all<PERM_B32>(d);
  return rc;
}
