// The table through which the host code (msm_hip.hip) reaches one curve's kernels and host arithmetic.  Every curve is compiled as its
// own translation unit (curve_<name>.hip: csrc/curve_unit.h instantiated with that curve's constants; BN254's unit lives in msm_hip.hip
// itself, which also takes the field-independent kernels and the layout constants from it) so that the units build in parallel; a unit
// hands its table over through one accessor.
#pragma once
#include <cstddef>
#include <cstdint>

// What differs between the curves: the kernels that do field arithmetic, and the host's window combine.  A context holds one.
struct CurveOps {
  void (*convert_points)(const uint32_t*, uint32_t*, size_t, uint32_t, uint32_t*);
  void (*precompute_tables)(uint32_t*, size_t, size_t, int, int, int);
  void (*endo_points)(uint32_t*, size_t, size_t, size_t);
  // k_count<C, 4, true> for C = 12 / 14 / 16: the first sort pass of endomorphism launches, which splits the scalars itself (csrc/glv.h)
  void (*count_split[3])(const uint32_t*, size_t, uint32_t, uint32_t, int, int, int, size_t, uint32_t*, uint16_t*, int, uint64_t*, uint32_t*, uint32_t*, size_t);
  void (*scalars_from_mont256)(const uint32_t*, uint32_t*, size_t, uint32_t*);
  void (*smvp_chunks)(const uint32_t*, const uint32_t*, const uint32_t*, size_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*,
                      uint32_t*, uint32_t);
  void (*smvp_stitch)(const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
  void (*smvp_stitch_big)(const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t);
  void (*rowcol_4_8)(const uint32_t*, uint32_t*, uint32_t*);
  void (*rowcol_2_8)(const uint32_t*, uint32_t*, uint32_t*);
  void (*rowcol_3_8)(const uint32_t*, uint32_t*, uint32_t*);
  void (*rowcol_4_6)(const uint32_t*, uint32_t*, uint32_t*);
  void (*rowcol_2_6)(const uint32_t*, uint32_t*, uint32_t*);
  void (*rowcol_2_4)(const uint32_t*, uint32_t*, uint32_t*);
  void (*bpr_w256)(const uint32_t*, const uint32_t*, uint32_t*, int);
  void (*bpr_final)(const uint32_t*, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int);
  void (*bpr_planes)(const uint32_t*, const uint32_t*, uint32_t*, int, uint32_t*, uint32_t*, uint32_t*);
  void (*bpr_planes_xyzz)(const uint32_t*, const uint32_t*, uint32_t*, int, uint32_t*, uint32_t*, uint32_t*);
  void (*bpr_final_planes)(const uint32_t*, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int);
  bool use_w256;    // the narrow reduce tail fits a workgroup's LDS (k_bpr_w256); else k_bpr_planes<true> + k_bpr_final_planes
  int coord_words;  // 32-bit words per coordinate on the wire: 8 (254 / 255-bit fields) or 12 (BLS12-381); point = 2, Jacobian record = 3 of them
  int rec_words, xyzz_words;  // device bucket record / scratch record sizes (words)
  bool glv;         // MSM_HIP_BASES_ENDOMORPHISM available
  void (*sample_scalars)(uint64_t, size_t, uint32_t*);
  void (*sample_points)(uint64_t, size_t, uint32_t*);
  void (*export_buckets)(const uint32_t*, uint32_t*, size_t);
  void (*test_fq)(int, const uint32_t*, const uint32_t*, uint32_t*, size_t);
  void (*test_g1)(int, const uint32_t*, const uint32_t*, uint32_t*, size_t);
  void (*test_g1_mul_u32)(const uint32_t*, const uint32_t*, uint32_t*, size_t);
  bool (*combine_windows)(const uint8_t*, int, int, uint8_t*);
  bool (*window_from_planes)(const uint8_t*, uint8_t*);
  bool (*combine_wide)(const uint8_t*, const uint8_t*, int, uint8_t*);
  bool (*combine_wide_pairs)(const uint8_t*, int, uint8_t*);
  int (*to_affine64)(const uint8_t*, uint8_t*);
};
#define MSM_CURVE_OPS(K, F)                                                                                                              \
  {K::k_convert_points, K::k_precompute_tables, K::k_endo_points, {K::k_count<12, 4, true>, K::k_count<14, 4, true>, K::k_count<16, 4, true>}, K::k_scalars_from_mont256, K::k_smvp_chunks, K::k_smvp_stitch, K::k_smvp_stitch_big,       \
   K::k_bpr_rowcol<4, 8>, K::k_bpr_rowcol<2, 8>, K::k_bpr_rowcol<3, 8>, K::k_bpr_rowcol<4, 6>, K::k_bpr_rowcol<2, 6>, K::k_bpr_rowcol<2, 4>, \
   K::k_bpr_w256, K::k_bpr_final, K::k_bpr_planes<false>, K::k_bpr_planes<true>, K::k_bpr_final_planes, K::BPR_USE_W256, K::CW, K::REC_WORDS, K::XYZZ_WORDS, F::GLV_SUPPORTED, K::k_sample_scalars, K::k_sample_points, K::k_export_buckets, K::k_test_fq, K::k_test_g1,                \
   K::k_test_g1_mul_u32, F::host::combine_windows, F::host::window_from_planes, F::host::combine_wide, F::host::combine_wide_pairs, F::host::to_affine64}

// A G2 unit (coordinates in Fq2, csrc/fq2.h): the same table (its point sampler draws multiples of the subgroup's generator)
#define MSM_CURVE_OPS_FQ2(K, F)                                                                                                          \
  {K::k_convert_points, K::k_precompute_tables, K::k_endo_points, {K::k_count<12, 4, true>, K::k_count<14, 4, true>, K::k_count<16, 4, true>}, K::k_scalars_from_mont256, K::k_smvp_chunks, K::k_smvp_stitch, K::k_smvp_stitch_big,       \
   K::k_bpr_rowcol<4, 8>, K::k_bpr_rowcol<2, 8>, K::k_bpr_rowcol<3, 8>, K::k_bpr_rowcol<4, 6>, K::k_bpr_rowcol<2, 6>, K::k_bpr_rowcol<2, 4>, \
   K::k_bpr_w256, K::k_bpr_final, K::k_bpr_planes<false>, K::k_bpr_planes<true>, K::k_bpr_final_planes, K::BPR_USE_W256, K::CW, K::REC_WORDS, K::XYZZ_WORDS, F::GLV_SUPPORTED, K::k_sample_scalars, K::k_sample_points, K::k_export_buckets, K::k_test_fq, K::k_test_g1,                \
   K::k_test_g1_mul_u32, F::host::combine_windows, F::host::window_from_planes, F::host::combine_wide, F::host::combine_wide_pairs, F::host::to_affine64}

// accessors of the separately compiled units (hidden: not part of the C ABI)
extern "C" {
__attribute__((visibility("hidden"))) const CurveOps* msm_hip_curve_ops_grumpkin(void);
__attribute__((visibility("hidden"))) const CurveOps* msm_hip_curve_ops_pallas(void);
__attribute__((visibility("hidden"))) const CurveOps* msm_hip_curve_ops_vesta(void);
__attribute__((visibility("hidden"))) const CurveOps* msm_hip_curve_ops_bls12_381(void);
__attribute__((visibility("hidden"))) const CurveOps* msm_hip_curve_ops_bn254_g2(void);
__attribute__((visibility("hidden"))) const CurveOps* msm_hip_curve_ops_bls12_381_g2(void);
}
