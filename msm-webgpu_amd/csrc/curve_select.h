// Which curve the arithmetic headers (fq29.h, fq29_asm.h, g1.h, host_g1.h, msm_kernels.h) are being instantiated for.
//
// The engine supports four curves of the form y^2 = x^3 + b over a prime field of at most 255 bits: BN254 G1 (the reference's curve),
// Grumpkin, its cycle partner (base fields agreeing in their top 128 bits: 2^261 / p = 169 for both), and the Pasta cycle,
// Pallas and Vesta (255-bit moduli: 2^261 / p = 127, still above what the group formulas need -- g1.h).  The headers are
// written once; msm_hip.hip includes them once per curve ("curve unit", csrc/curve_unit.h) with
//     MSM_FIELD_NS          namespace of the field / curve arithmetic and its constants      (bn254 | grumpkin | pallas | vesta)
//     MSM_KERNEL_NS         namespace of the kernels                                          (msmk  | msmk_<curve>)
//     MSM_CURVE_CONSTANTS   the generated constants header                                    (tools/gen_constants.py)
// Included stand-alone (host test harness, micro-benchmarks) they instantiate BN254.
#ifndef MSM_FIELD_NS
#define MSM_FIELD_NS bn254
#define MSM_KERNEL_NS msmk
#define MSM_CURVE_CONSTANTS "bn254_constants.h"
#endif
