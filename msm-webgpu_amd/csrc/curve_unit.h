// One curve's worth of device + host arithmetic and kernels: define MSM_FIELD_NS, MSM_KERNEL_NS and MSM_CURVE_CONSTANTS, include
// this file, undefine them (csrc/curve_select.h).  No include guard on purpose: msm_hip.hip includes it once per curve.
#define MSM_CURVE_UNIT 1
#include MSM_CURVE_CONSTANTS
#include "fq29.h"
#include "g1.h"
#include "host_g1.h"
#include "glv.h"
#include "msm_kernels.h"
#undef MSM_CURVE_UNIT
