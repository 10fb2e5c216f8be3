// BLS12-381 G1 as its own translation unit of libmsm_hip.so (the reference lists other curves as future work, README.md):
// the arithmetic headers and the kernels instantiated with this curve's constants and limb layout -- 14 limbs of 28 bits for the 381-bit
// base field, 48-byte coordinates on the wire (csrc/curve_unit.h, csrc/curve_select.h) -- and the table through which msm_hip.hip
// reaches them (csrc/curve_ops.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/msm_hip.h"
#define MSM_FIELD_NS bls12_381
#define MSM_KERNEL_NS msmk_bls12_381
#define MSM_CURVE_CONSTANTS "bls12_381_constants.h"
#define MSM_FQ_ASM_HEADER "fq28x14_asm.h"
#include "curve_unit.h"
#include "curve_ops.h"

extern "C" const CurveOps* msm_hip_curve_ops_bls12_381(void) {
  static const CurveOps ops = MSM_CURVE_OPS(msmk_bls12_381, bls12_381);
  return &ops;
}
