// BLS12-381 G2 as its own translation unit of libmsm_hip.so (SURVEY.md 8f-4 "other curves / G2"): the twist y^2 = x^3 + 4 (1 + u) over
// Fq2 = Fq[u] / (u^2 + 1), scalars modulo the same r as BLS12-381 G1.  As csrc/curve_bn254_g2.hip, on the 14 x 28-bit prime field: 28 limbs
// per coordinate, 96-byte coordinates c0 || c1 on the wire, 192-byte points, 288-byte Jacobian records.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/msm_hip.h"
#define MSM_CURVE_UNIT 1
// the prime field Fq (14 x 28-bit limbs, its generated multipliers)
#define MSM_FIELD_NS bls12_381_g2_fp
#define MSM_FQ_ASM_HEADER "fq28x14_asm.h"
#include "bls12_381_constants.h"
#include "fq29.h"
#undef MSM_FIELD_NS
// the coordinate field Fq2 and everything above it
#define MSM_FQ2 1
#define MSM_G1_OUTLINE 1  // g1_add / g1_double as calls: see csrc/g1.h
#define MSM_BASE_NS bls12_381_g2_fp
#define MSM_FIELD_NS bls12_381_g2
#define MSM_KERNEL_NS msmk_bls12_381_g2
#include "bls12_381_g2_constants.h"
#include "fq2.h"
#include "g1.h"
#include "host_g1.h"
#include "glv.h"
#include "msm_kernels.h"
#undef MSM_CURVE_UNIT
#include "curve_ops.h"

extern "C" const CurveOps* msm_hip_curve_ops_bls12_381_g2(void) {
  static const CurveOps ops = MSM_CURVE_OPS_FQ2(msmk_bls12_381_g2, bls12_381_g2);
  return &ops;
}
