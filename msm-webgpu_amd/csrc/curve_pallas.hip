// Pallas as its own translation unit of libmsm_hip.so: the arithmetic headers and the kernels instantiated with this curve's constants
// (csrc/curve_unit.h, csrc/curve_select.h) and the table through which msm_hip.hip reaches them (csrc/curve_ops.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/msm_hip.h"
#define MSM_FIELD_NS pallas
#define MSM_KERNEL_NS msmk_pallas
#define MSM_CURVE_CONSTANTS "pallas_constants.h"
#include "curve_unit.h"
#include "curve_ops.h"

extern "C" const CurveOps* msm_hip_curve_ops_pallas(void) {
  static const CurveOps ops = MSM_CURVE_OPS(msmk_pallas, pallas);
  return &ops;
}
