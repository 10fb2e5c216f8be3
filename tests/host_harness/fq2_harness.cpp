// Host build (g++ -DFQ_CHECK) of the G2 unit's arithmetic -- the prime field, csrc/fq2.h on top of it, g1.h over Fq2 -- set up exactly as
// csrc/curve_bn254_g2.hip does, behind the entry points of fq29_harness.cpp.  Test-only: not part of libmsm_hip.so.
#define MSM_CURVE_UNIT 1
#define MSM_FIELD_NS bn254_g2_fp
#include "bn254_constants.h"
#include "fq29.h"
#undef MSM_FIELD_NS
#define MSM_FQ2 1
#define MSM_BASE_NS bn254_g2_fp
#define MSM_FIELD_NS bn254_g2
#include "bn254_g2_constants.h"
#include "fq2.h"
#include "g1.h"
#define HARNESS_FIELD_NS bn254_g2
#define HARNESS_PRELUDE_DONE 1
#include "fq29_harness.cpp"
