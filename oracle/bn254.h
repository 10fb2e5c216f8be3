/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc) of the BN254 G1 MSM hot path of ICME-Lab/msm-webgpu, used ONLY as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the product
 * (msm-webgpu_amd/) links, imports or calls this library.
 *
 * Parity status: the reference holds no golden MSM vectors and cannot be built here (Rust + WGSL, no
 * toolchain; halo2curves 0.9.0 is an un-vendored Cargo dependency, Cargo.lock:500-503), so MSM-output
 * parity is UNPINNED BY REFERENCE DATA.  It is pinned (a) mathematically: the MSM result is a unique
 * group element whose canonical affine encoding is implementation independent, (b) by the reference's
 * own known-answer constants (src/cuzk/utils.rs:439-451, src/naive/utils/bigint.rs:83-93,
 * src/cuzk/msm.rs:39), (c) by cross-checking against the independent pure-Python model
 * oracle/bn254_ref.py on the committed vectors in tests/golden/.
 *
 * What each group restates (paths relative to /root/reference):
 *   oracle_msm_*            cpu_msm -> halo2curves::msm::msm_best        src/lib.rs:45-47 (algorithm: halo2curves 0.9.0
 *                           src/msm.rs, Booth-recoded windows c = ceil(ln n), buckets, summation by parts)
 *   oracle_decompose_*      decompose_scalars_signed                      src/cuzk/test/utils.rs:121-161
 *   oracle_transpose        cpu_transpose                                 src/cuzk/test/utils.rs:61-118
 *   oracle_smvp_signed      cpu_smvp_signed                               src/cuzk/test/utils.rs:166-219
 *   oracle_*_bucket_reduction  serial / running-sum / parallel (1,2)      src/cuzk/test/utils.rs:222-338
 *   oracle_horner           host finalisation                             src/cuzk/msm.rs:411-416
 *   oracle_points/scalars_* wire format                                   src/lib.rs:50-65, src/cuzk/utils.rs:10-21
 * Builds: BN254 G1 (default), -DORACLE_GRUMPKIN / _PALLAS / _VESTA / _BLS12_381 (other G1 curves), and -DORACLE_G2 (with or without
 * -DORACLE_BLS12_381): the same restatement over G2 of BN254 / BLS12-381, coordinates in Fq2 = Fq[u] / (u^2 + 1) (bn254.c).
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(ORACLE_BLS12_381)
#define ONL 6 /* 64-bit limbs of a base-field element: BLS12-381's p has 381 bits (48-byte coordinates, 96-byte points, 144-byte Jacobian records:
                 every "32 B" / "64 B" / "96 B" below reads 48 / 96 / 144 for this build; scalars stay 32 B) */
#else
#define ONL 4
#endif
#if defined(ORACLE_G2) /* the G2 builds: coordinates in Fq2 = Fq[u] / (u^2 + 1), c0 || c1 on the wire (64 B; 96 B with ORACLE_BLS12_381) */
typedef struct { uint64_t l[ONL]; } ofp;        /* prime-field element, Montgomery form, R = 2^(64 ONL) */
typedef struct { ofp c0, c1; } ofq;             /* c0 + c1 u */
#else
typedef struct { uint64_t l[ONL]; } ofq;        /* Fq element, Montgomery form, R = 2^(64 ONL) */
#endif
typedef struct { ofq x, y, z; } og1;            /* Jacobian; z == 0 <=> identity (ec.template.wgsl:4) */

/* ---- field / point op hooks (≙ tests/field.rs, tests/point.rs) : all I/O canonical little-endian ---- */
/* op: 0 add, 1 sub, 2 mul, 3 sqr(a), 4 neg(a), 5 inv(a)   a,b,out: n x 32 B */
void oracle_fq_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
/* a, b: n x 96 B Jacobian x||y||z (z=0 identity); op: 0 add, 1 double(a), 2 negate(a); out n x 96 B Jacobian */
void oracle_g1_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
/* out[i] = k[i] * P[i];  P: n x 64 B affine, k: n x 32 B scalars, out: n x 96 B Jacobian */
void oracle_g1_scalar_mul(const uint8_t* p_xy, const uint8_t* k, uint8_t* out, size_t n);
/* 96 B Jacobian -> 64 B canonical affine (64 zero bytes for the identity); returns 1 if identity */
int oracle_g1_to_affine64(const uint8_t* xyz, uint8_t* out);
/* 1 if every point satisfies y^2 = x^3 + 3 with canonical coordinates */
int oracle_points_on_curve(const uint8_t* xy, size_t n);

/* ---- MSM (≙ cpu_msm, src/lib.rs:45-47).  out: 96 B Jacobian canonical LE.  returns 0 ---- */
int oracle_msm_bn254_g1(const uint8_t* xy, const uint8_t* scalars, size_t n, uint8_t* out_xyz);
/* same, split over n_threads pthreads by point ranges (plonky2_maybe_rayon with rayon would do this; the
 * reference lockfile is serial, Cargo.lock:961-964) */
int oracle_msm_bn254_g1_mt(const uint8_t* xy, const uint8_t* scalars, size_t n, int n_threads, uint8_t* out_xyz);

/* ---- cuZK stage models ---- */
/* digits[w*n + i] = biased signed digit in [0, 2^c); returns -1 if a final carry occurs */
int oracle_decompose_scalars_signed(const uint8_t* scalars, size_t n, int num_words, int word_size, int32_t* digits);
/* one window: col_ptr[num_columns+1], val_idxs[n] */
void oracle_transpose(const int32_t* digits_w, size_t n, int num_columns, int32_t* col_ptr, int32_t* val_idxs);
/* one window: buckets[(num_columns/2)] x 96 B Jacobian canonical */
void oracle_smvp_signed(const int32_t* col_ptr, const int32_t* val_idxs, const uint8_t* xy, size_t n, int num_columns,
                        uint8_t* buckets_xyz);
/* kind: 0 serial (k * B[k]), 1 running-sum, 2 parallel (num_threads simulated, results summed) ; out 96 B */
void oracle_bucket_reduction(int kind, const uint8_t* buckets_xyz, int num_buckets, int num_threads, uint8_t* out_xyz);
/* parallel_bucket_reduction_1 + _2 kept separate: g_out/m_out num_threads x 96 B each */
void oracle_parallel_bucket_reduction_1(const uint8_t* buckets_xyz, int num_buckets, int num_threads, uint8_t* g_out,
                                        uint8_t* m_out);
void oracle_parallel_bucket_reduction_2(const uint8_t* g_in, const uint8_t* m_in, int num_buckets, int num_threads,
                                        uint8_t* out);
/* result = sum_w 2^(word_size*w) * S_w ; window_sums: num_words x 96 B */
void oracle_horner(const uint8_t* window_sums_xyz, int num_words, int word_size, uint8_t* out_xyz);
/* the whole cuZK pipeline on the CPU stage models (cf. tests/cuzk.rs:11-95) */
int oracle_msm_cuzk_model(const uint8_t* xy, const uint8_t* scalars, size_t n, int word_size, uint8_t* out_xyz);

/* ---- deterministic synthetic inputs (same definition as oracle/bn254_ref.py and the HIP samplers) ---- */
void oracle_sample_scalars(uint64_t seed, size_t first, size_t n, uint8_t* out32);
void oracle_sample_points(uint64_t seed, size_t first, size_t n, uint8_t* out64);

/* bytes of a coordinate on this build's wire: 32, or 48 for -DORACLE_BLS12_381 */
int oracle_coord_bytes(void);
/* constants for KAT tests: writes LE values (p, R^2 mod p, R mod p: oracle_coord_bytes() bytes each; r: 32 bytes) */
void oracle_constants(uint8_t* p, uint8_t r[32], uint8_t* r2_mod_p, uint8_t* one_mont, uint64_t* n0inv64);

#ifdef __cplusplus
}
#endif
#endif
