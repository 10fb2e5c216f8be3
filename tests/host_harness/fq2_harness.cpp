// Host build (g++ -DFQ_CHECK) of a G2 unit's arithmetic -- the prime field, csrc/fq2.h on top of it, g1.h over Fq2 -- set up exactly as
// csrc/curve_bn254_g2.hip does (-DHARNESS_G2_BLS12_381: as csrc/curve_bls12_381_g2.hip), behind the entry points of fq29_harness.cpp.
// Test-only: not part of libmsm_hip.so.
#define MSM_CURVE_UNIT 1
#ifdef HARNESS_G2_BLS12_381
#define MSM_FIELD_NS bls12_381_g2_fp
#include "bls12_381_constants.h"
#else
#define MSM_FIELD_NS bn254_g2_fp
#include "bn254_constants.h"
#endif
#include "fq29.h"
#undef MSM_FIELD_NS
#define MSM_FQ2 1
#ifdef HARNESS_G2_BLS12_381
#define MSM_BASE_NS bls12_381_g2_fp
#define MSM_FIELD_NS bls12_381_g2
#include "bls12_381_g2_constants.h"
#else
#define MSM_BASE_NS bn254_g2_fp
#define MSM_FIELD_NS bn254_g2
#include "bn254_g2_constants.h"
#endif
#include "fq2.h"
#include "g1.h"
#define HARNESS_FIELD_NS MSM_FIELD_NS
#define HARNESS_PRELUDE_DONE 1
#include "fq29_harness.cpp"
