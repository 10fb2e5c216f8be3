// BN254 G2 as its own translation unit of libmsm_hip.so (SURVEY.md 8f-4 "other curves / G2"; the reference lists other curves as future
// work, README.md): the twist y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1), scalars modulo the same r as G1.  The prime field is
// instantiated first, in a namespace of its own; csrc/fq2.h then provides the field interface of g1.h / msm_kernels.h / host_g1.h over Fq2
// (18 limbs per coordinate, 64-byte coordinates c0 || c1 on the wire, 128-byte points, 192-byte Jacobian records), and the same kernels
// are compiled against it.  csrc/curve_ops.h: the table through which msm_hip.hip reaches them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/msm_hip.h"
#define MSM_CURVE_UNIT 1
// the prime field Fq (9 x 29-bit limbs, the generated multipliers)
#define MSM_FIELD_NS bn254_g2_fp
#include "bn254_constants.h"
#include "fq29.h"
#undef MSM_FIELD_NS
// the coordinate field Fq2 and everything above it
#define MSM_FQ2 1
#define MSM_BASE_NS bn254_g2_fp
#define MSM_FIELD_NS bn254_g2
#define MSM_KERNEL_NS msmk_bn254_g2
#include "bn254_g2_constants.h"
#include "fq2.h"
#include "g1.h"
#include "host_g1.h"
#include "glv.h"
#include "msm_kernels.h"
#undef MSM_CURVE_UNIT
#include "curve_ops.h"

extern "C" const CurveOps* msm_hip_curve_ops_bn254_g2(void) {
  static const CurveOps ops = MSM_CURVE_OPS_FQ2(msmk_bn254_g2, bn254_g2);
  return &ops;
}
