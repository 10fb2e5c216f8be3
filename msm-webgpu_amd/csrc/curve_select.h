// Which curve the arithmetic headers (fq29.h, fq29_asm.h, g1.h, host_g1.h, msm_kernels.h) are being instantiated for.
//
// The engine supports two curves whose base fields agree in their top 128 bits (so that every limb / value bound of fq29.h and
// g1.h holds for both): BN254 G1 (the reference's curve) and Grumpkin, its cycle partner.  The headers are
// written once; msm_hip.hip includes them once per curve ("curve unit", csrc/curve_unit.h) with
//     MSM_FIELD_NS          namespace of the field / curve arithmetic and its constants      (bn254 | grumpkin)
//     MSM_KERNEL_NS         namespace of the kernels                                          (msmk  | msmk_grumpkin)
//     MSM_CURVE_CONSTANTS   the generated constants header                                    (tools/gen_constants.py)
// Included stand-alone (host test harness, micro-benchmarks) they instantiate BN254.
#ifndef MSM_FIELD_NS
#define MSM_FIELD_NS bn254
#define MSM_KERNEL_NS msmk
#define MSM_CURVE_CONSTANTS "bn254_constants.h"
#endif
