/* libmsm_hip -- C ABI of the MI355X-native BN254 G1 multi-scalar-multiplication engine.
 *
 * Drop-in boundary for the hot path of ICME-Lab/msm-webgpu (all reference paths relative to /root/reference):
 *
 *     pub async fn run_webgpu_msm<C: CurveAffine>(g: &[C], v: &[C::Scalar]) -> C::Curve      src/lib.rs:76-82
 *     pub async fn compute_msm<C: CurveAffine>(points: &[C], scalars: &[C::Scalar]) -> C::Curve   src/cuzk/msm.rs:75-417
 *
 * The reference has no FFI layer; a Rust maintainer binds these entry points with `extern "C"` and keeps the two
 * signatures above (INTEGRATION.md shows the shim).  Wire formats are exactly what the reference's own helpers emit:
 *
 *     points  : n x 64 B, x || y, each coordinate the canonical (non-Montgomery) integer in [0, p), little-endian
 *               = points_to_bytes()   src/lib.rs:55-65  (the point at infinity is not representable, lib.rs:58 panics)
 *     scalars : n x 32 B, canonical integer in [0, r), little-endian
 *               = scalars_to_bytes()  src/lib.rs:50-52, field_to_bytes src/cuzk/utils.rs:10-14
 *     result  : 96 B Jacobian x || y || z, canonical little-endian, z = 0 <=> identity
 *               = what C::Curve::new_jacobian consumes at src/cuzk/msm.rs:394
 *
 * Conventions: every function returns MSM_HIP_OK (0) or a negative error; nothing aborts or throws across the ABI
 * (the reference panics instead: src/cuzk/gpu.rs:22,51, src/lib.rs:58, src/cuzk/msm.rs:399, src/cuzk/utils.rs:20).
 * The caller owns all host buffers.  A context is used by one host thread at a time.  There is no CPU fallback:
 * with no usable HIP device every entry point fails with MSM_HIP_ERR_NO_DEVICE.
 */
#ifndef MSM_HIP_H
#define MSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSM_HIP_OK 0
#define MSM_HIP_ERR_NO_DEVICE (-1)     /* no HIP device / device init failed            (≙ gpu.rs:22,51 expect) */
#define MSM_HIP_ERR_INVALID_ARG (-2)   /* null pointer, n mismatch, bad window range                             */
#define MSM_HIP_ERR_OUT_OF_MEMORY (-3) /* hipMalloc failed                                                       */
#define MSM_HIP_ERR_NONCANONICAL (-4)  /* a coordinate >= p or a scalar whose recode overflows (≙ utils.rs:20)   */
#define MSM_HIP_ERR_NOT_ON_CURVE (-5)  /* only with MSM_HIP_CHECK_ON_CURVE                                      */
#define MSM_HIP_ERR_NO_BASES (-6)      /* run called before set_bases                                            */
#define MSM_HIP_ERR_HIP (-7)           /* a HIP runtime call failed; msm_hip_last_hip_error() has the code        */
#define MSM_HIP_ERR_SLOT_BUSY (-8)     /* launch into a result slot whose previous MSM was not finished / synced   */

#define MSM_HIP_NUM_WINDOWS 16         /* num_subtasks = ceil(256 / 16)                       src/cuzk/msm.rs:82 */
#define MSM_HIP_WINDOW_BITS 16         /* chunk_size                                           src/cuzk/msm.rs:79 */
#define MSM_HIP_MAX_LOCAL_WINDOWS 64   /* (scalar vectors) x (windows of each) one launch may carry                */
#define MSM_HIP_NUM_SLOTS 4            /* asynchronous result slots per context                                 */
#define MSM_HIP_BUCKETS_PER_WINDOW 32768 /* 2^(c-1) signed buckets                             src/cuzk/msm.rs:191 */

/* flags for msm_hip_set_bases_* */
#define MSM_HIP_CHECK_ON_CURVE 1u      /* verify y^2 = x^3 + 3 for every base (one extra square + cube per point) */
#define MSM_HIP_BASES_MONT256 2u       /* coordinates are x * 2^256 mod p, little-endian (the in-memory form of a 4 x 64-bit
                                          Montgomery field library with R = 2^256 -- what halo2curves is believed to hold;
                                          its layout could not be verified here, so the canonical format is the contract and
                                          this one an opt-in that saves the caller two from-Montgomery conversions per point) */

#define MSM_HIP_BASES_PRECOMPUTE 4u    /* fixed-base tables (reference README.md "Future work"): also store 2^(16 w) P_i
                                          for w = 1 .. 15 (16 x the base memory: 1 GiB at 2^20 points; at most 2^24 points).  Whole-MSM
                                          entry points (run, launch / finish, batch) then put all 16 windows of an MSM into ONE bucket
                                          set: one stitch + bucket reduce instead of 16, no window combine, up to 64 MSMs per launch.
                                          Same result.  The window-sharding entry points ignore the tables (table 0 is the plain set). */
#define MSM_HIP_BASES_ENDOMORPHISM 8u  /* curve endomorphism (SURVEY.md 8f-3 "GLV endomorphism ... to halve scalar length"; the reference
                                          uses full-length scalars, src/cuzk/msm.rs:79-82): also store phi(P_i) = (beta x_i, y_i) = lambda P_i
                                          (2 x the base memory; at most 2^27 points).  Whole-MSM entry points then split every scalar
                                          k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^127, on the device and run the MSM over the 2n points
                                          with HALF the windows (8 instead of 16 at 16 bits): the same bucket additions, half the buckets
                                          to stitch / reduce and half the window sums to combine.  Same result.  Not combinable with
                                          MSM_HIP_BASES_PRECOMPUTE; the window-sharding entry points ignore it (records 0 .. n-1 are the
                                          plain set). */

#define MSM_HIP_BASES_PRECOMPUTE_WIDE 32u /* wide fixed-base tables (round 4): store 2^(C w) P_i for signed digits of C bits and recode every scalar into
                                          ceil(255 / C) of them -- 15 bucket additions per point at C = 17, 14 at 19, 13 at 20, instead of the reference's
                                          16 (src/cuzk/msm.rs:79-82: chunk_size 16) -- into ONE bucket set of 2^(C-1) slots, run as 2^(C-16) "virtual
                                          windows" of 2^15 slots.  C by the number of bases: 16 up to 2^16 points (the 16-bit tables behind this mode's sort),
                                          17 up to 2^20 (19 on BLS12-381, whose scalar field 15 digits of 17 bits cannot hold), 20 beyond;
                                          msm_hip_set_wide_bits overrides.
                                          The top table is 2^(C (T-1) - t) P_i and the top digit is used shifted by t, so that it spreads over the bucket
                                          set (t from the scalar field's modulus; exact for any point).  For fixed bases: over the endomorphism mode +44 ... 58 % at
                                          2^14, +26 ... 50 % at 2^16, +20 % at 2^18 (grouped launches; 50 / 20 timed steps), +4 % at 2^20, +11 % at 2^22, +18 % at 2^24; 15 x (13 x) the
                                          base memory, at most 2^24 points; up to 12 (at 20 bits: 1) whole MSMs per launch; sort arrays sized for a skewed
                                          vector (2^(C-16) x T x n entries per MSM: 0.3 GiB at 2^20, 31 GiB at 2^24).
                                          Whole-MSM entry points (run, launch / finish, batch) and -- round 5 -- shares of the virtual windows for the
                                          window-sharded / multi-GPU paths (msm_hip_launch_vwindows_batch_device, msm_hip_mgpu_*); the 16-bit
                                          window-sharding entry points ignore the tables (table 0 is the plain set).  Same result for every scalar below the scalar field's modulus (a top digit that
                                          does not fit after its shift is MSM_HIP_ERR_NONCANONICAL).  Not combinable with the other two modes. */

#define MSM_HIP_BASES_PLAIN 16u        /* hold the n bases only and run the reference's exact shape -- 16 windows of full-length scalars over n points
                                          (src/cuzk/msm.rs:79-82).  WITHOUT this flag, MSM_HIP_BASES_PRECOMPUTE or MSM_HIP_BASES_ENDOMORPHISM a
                                          base set takes the fastest mode the curve has (round 4: the drop-in default is the headline's mode):
                                          the endomorphism mode on the curves of prime order that have one (BN254 G1, Grumpkin, Pallas, Vesta; up to
                                          2^27 points), else the plain shape (on a curve with a cofactor -- BLS12-381, the G2 twists -- the mode
                                          needs bases of order r and therefore stays an explicit choice).
                                          Same group element either way; what the flag decides is memory (n or 2n records) and speed. */

typedef struct msm_hip_ctx msm_hip_ctx;

/* ---- context: replaces get_adapter/get_device + per-call buffer/pipeline creation (src/cuzk/gpu.rs:11-54,
 *      src/cuzk/msm.rs:88-94).  Persistent: one stream, pooled device buffers, resident bases. ---- */
/* Every entry point runs on the context's device and restores the caller's current HIP device before it returns.
 * Result slots (bucket and piece arrays, events) are set up by the first launch that uses them, so a context that runs one
 * MSM at a time -- the reference's call shape, src/cuzk/msm.rs:75-94 -- holds one slot's memory, not MSM_HIP_NUM_SLOTS. */
int msm_hip_ctx_create(msm_hip_ctx** out, int device_id);
void msm_hip_ctx_destroy(msm_hip_ctx* ctx);

/* ---- other curves (the reference lists them as future work, README.md, and is hard-wired to BN254's Fq,
 *      src/cuzk/msm.rs:37-43).  The curve is a property of the context: every entry point of this header works on the curve its
 *      context was created for (round 5: the entry points carry curve-neutral names; the `_bn254` names of rounds 1 - 4 -- the reference's only
 *      instantiation -- remain as aliases, at the end of this header).  Supported besides BN254 G1:
 *      Grumpkin, BN254's cycle partner (y^2 = x^3 - 17 over BN254's scalar field r; scalars modulo BN254's base field p) -- its
 *      base field agrees with BN254's in the top 128 bits, so the same 9 x 29-bit lazy-limb arithmetic and the same kernels serve
 *      both, instantiated once per curve (csrc/curve_select.h); and the Pasta cycle, Pallas and Vesta (y^2 = x^3 + 5 over the
 *      255-bit p with q points, resp. over q with p points; generators (-1, 2)) -- the reference's own dead second curve is Pallas
 *      (src/naive/wgsl/pallas) -- whose moduli leave 2^261 / p = 127 instead of 169: enough for the formulas.
 *      Wire formats are the same with the curve's own moduli. ---- */
#define MSM_HIP_CURVE_BN254_G1 0
#define MSM_HIP_CURVE_GRUMPKIN 1
#define MSM_HIP_CURVE_PALLAS 2
#define MSM_HIP_CURVE_VESTA 3
#define MSM_HIP_CURVE_BLS12_381 4 /* G1: y^2 = x^3 + 4 over the 381-bit p, scalars modulo the 255-bit r.  Coordinates are 48 bytes on this curve's wire:
                                     points n x 96 B (x || y), results and window sums 144 B Jacobian records (x || y || z), scalars 32 B as everywhere;
                                     every `[96]` / `[64]` / "x 96 B" of this header reads 144 / 96 for such a context.  Device arithmetic: 14 limbs of
                                     28 bits (csrc/curve_bls12_381.hip).  All modes (window sizes, fixed-base tables, endomorphism, shards, batches). */
#define MSM_HIP_CURVE_BN254_G2 5 /* G2 of BN254: the twist y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1), scalars modulo the same r as curve 0.
                                    A coordinate is an Fq2 element c0 || c1, each 32 B canonical little-endian: 64 bytes; points n x 128 B (x || y), results
                                    and window sums 192 B Jacobian records, scalars 32 B; every `[96]` / `[64]` / "x 96 B" of this header reads 192 / 128.
                                    Device arithmetic: csrc/fq2.h on the 9 x 29-bit prime field (csrc/curve_bn254_g2.hip).  Window sizes, window shards,
                                    batches, grouped launches, fixed-base tables, Montgomery-form inputs, the multi-GPU calls: as for curve 0.
                                    MSM_HIP_BASES_ENDOMORPHISM (round 4): the twist has j = 0 like the curve, so (beta x, y) with beta in the PRIME field is the
                                    multiplication by the same lambda on G2 -- same split, half-length scalars over 2n points; the bases must have order r (G2
                                    proper), which is why the mode is never the default here.  msm_hip_sample_points_device draws P_i = (a + i b) G for seeded a, b
                                    and the standard generator G of the subgroup (points of order r; oracle/bn254_g2_ref.py: sample_points). */
#define MSM_HIP_CURVE_BLS12_381_G2 6 /* G2 of BLS12-381: the twist y^2 = x^3 + 4 (1 + u) over Fq2, scalars modulo the same r as curve 4.  As curve 5 with 48-byte
                                        components: coordinates 96 B (c0 || c1), points 192 B, results and window sums 288 B Jacobian records.  The same
                                        options are unavailable.  Inputs are expected in the order-r subgroup (as every valid G2 point is); points outside
                                        it are summed as the integers their scalars are, like on curve 4. */
#define MSM_HIP_NUM_CURVES 7
int msm_hip_ctx_create_curve(msm_hip_ctx** out, int device_id, int curve);
int msm_hip_ctx_curve(const msm_hip_ctx* ctx);
/* the context-free host helpers for a given curve (msm_hip_combine_windows_bn254 / msm_hip_g1_to_affine_bn254 are curve 0) */
int msm_hip_combine_windows_curve(int curve, const uint8_t* window_sums_host, int num_windows, uint8_t out_xyz[96]);
int msm_hip_g1_to_affine_curve(int curve, const uint8_t xyz[96], uint8_t out_xy[64]);

/* ---- ordering contract for DEVICE inputs.  The engine works on its own non-blocking HIP streams, which are not ordered
 *      with the caller's streams (nor with the null stream).  Device scalars / bases handed to any *_device_* entry point
 *      must therefore be complete before the call -- or the caller names the stream that produces them:
 *      msm_hip_wait_stream makes the engine's main stream wait (on the device, no host block) for everything enqueued so
 *      far on `producer_stream` (a hipStream_t; NULL = the legacy default stream); call it just before the launch.
 *      Input buffers must stay alive and unmodified until the launch's slot is collected (finish / slot_sync). ---- */
int msm_hip_wait_stream(msm_hip_ctx* ctx, void* producer_stream);

/* ---- bases: upload + convert to the device's Montgomery form once (≙ the point half of the decompose shader,
 *      src/cuzk/wgsl/cuzk/decompose_scalars.template.wgsl:41-70, launched at src/cuzk/msm.rs:441-524) ---- */
int msm_hip_set_bases(msm_hip_ctx* ctx, const uint8_t* xy_host, size_t n, uint32_t flags);
/* same, bytes already in device memory (plain device pointer, e.g. a torch tensor's data_ptr) */
int msm_hip_set_bases_device(msm_hip_ctx* ctx, const void* xy_dev, size_t n, uint32_t flags);

/* ---- scalar format of all following runs of this context: canonical little-endian integers (default, = scalars_to_bytes,
 *      src/lib.rs:50-52), or s * 2^256 mod r, little-endian -- the in-memory limbs of a 4 x 64-bit Montgomery library with
 *      R = 2^256 (what halo2curves is believed to hold; unverified here, hence opt-in).  The second form spares the caller
 *      one from-Montgomery conversion per scalar (`to_repr`, ~40 ms of one CPU core per 2^20 scalars -- 25 x the GPU time of
 *      the whole MSM); the device converts them in a pre-pass (one extra read and write of the scalars). ---- */
#define MSM_HIP_SCALARS_CANONICAL 0u
#define MSM_HIP_SCALARS_MONT256 1u
int msm_hip_set_scalar_format(msm_hip_ctx* ctx, uint32_t format);

/* ---- window size (the reference hard-codes chunk_size = 16 for n >= 2^16, src/cuzk/msm.rs:79-82).
 *      WHOLE-MSM entry points (run, launch, finish, batch) pick the signed-digit window from n: 12 bits (22 windows of 2^11 buckets)
 *      up to 2^12 points, 16 bits (16 windows of 2^15 buckets) beyond -- the measured optimum on MI355X for one MSM per launch; launches
 *      that carry several whole MSMs (the batch entry points) use 14 bits (19 windows of 2^13 buckets) up to 2^16 points.
 *      msm_hip_set_window_bits fixes the size (12, 14 or 16; 0 = by n).
 *      The result is the same group element for every window size.  The window-sharding entry points below (w_begin / w_end) and
 *      msm_hip_combine_windows_bn254 always use the reference's 16-bit windows. ---- */
int msm_hip_set_window_bits(msm_hip_ctx* ctx, int bits);
int msm_hip_window_config(int bits, int* num_windows, int* buckets_per_window); /* host-only: the shape of a window size */
/* digit width of the wide fixed-base tables the NEXT msm_hip_set_bases_*(…, MSM_HIP_BASES_PRECOMPUTE_WIDE) builds: 16 .. 20, 0 = by the number
 * of bases and the curve (see the flag).  A width that cannot hold the curve's scalars (17 on BLS12-381) makes that call fail with
 * MSM_HIP_ERR_INVALID_ARG.  msm_hip_wide_bits: the width of the resident tables (0: none). */
int msm_hip_set_wide_bits(msm_hip_ctx* ctx, int bits);
int msm_hip_wide_bits(const msm_hip_ctx* ctx);
/* host-only: the shape of the wide tables for `curve` (MSM_HIP_CURVE_*) at `bits` (0: the width picked for n bases): digit width, number of
 * tables (= bucket additions per point), virtual windows of 2^15 slots, and the shift of the top digit.  MSM_HIP_ERR_INVALID_ARG: that width
 * cannot hold the curve's scalars. */
int msm_hip_wide_config(int curve, int bits, size_t n, int* digit_bits, int* tables, int* virtual_windows, int* top_shift);
int msm_hip_last_window_bits(msm_hip_ctx* ctx);                                 /* window size of the last launch       */
int msm_hip_endomorphism_window_count(int bits); /* host-only: windows of a 127-bit half (MSM_HIP_BASES_ENDOMORPHISM): 8 / 10 / 11 */
int msm_hip_uses_endomorphism(const msm_hip_ctx* ctx); /* 1: the resident bases were set with MSM_HIP_BASES_ENDOMORPHISM */
/* how many whole MSMs of n points the batch entry points put through ONE launch (what msm_hip_launch_windows_batch_device
 * with w_begin = 0, w_end = 16, window_sums_dev = NULL should be given for best throughput): 1 from 2^20 points up, at most
 * MSM_HIP_MAX_LOCAL_WINDOWS / (windows of the size picked for n) below */
int msm_hip_batch_group_size(msm_hip_ctx* ctx, size_t n);

/* ---- run: sum_i scalars[i] * bases[i] over the first n bases (n <= number of bases set).
 *      ≙ compute_msm stages 1-5, src/cuzk/msm.rs:96-416 (decompose, transpose, SMVP, bucket reduce, Horner).
 *      Host scalars, from 2^20 points on (round 5; not with fixed-base tables, not while the caller has launches of its own in slots 1 - 2): the call is
 *      the sum of 2 (from 2^22: 3) sub-MSMs over ranges of the points, one result slot each, so that a range's scalars cross the host link while
 *      the previous range is accumulated; the results are added on the host.  Same group element. ---- */
int msm_hip_run(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]);
int msm_hip_run_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, uint8_t out_xyz[96]);
/* asynchronous halves of run_device: `launch` enqueues all device work of one MSM into result slot `slot` (0 .. MSM_HIP_NUM_SLOTS-1)
 * and returns; `finish` waits for that slot and performs the host finalisation (src/cuzk/msm.rs:391-416).
 * Lets a caller overlap the host Horner of MSM i with the device work of MSM i+1. */
int msm_hip_launch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int slot);
int msm_hip_finish(msm_hip_ctx* ctx, int slot, uint8_t out_xyz[96]);
/* `launch` with the scalars in HOST memory (msm_hip_run = this + finish on slot 0): they are copied into the slot's own
 * device staging buffer on a separate copy stream, so the copy of MSM i+1 overlaps the device work of MSM i when the caller
 * alternates slots.  Pageable memory: returns once the bytes have left the caller's buffer.  Pinned memory (hipHostMalloc /
 * hipHostRegister): returns at once and the buffer must stay untouched until the slot is collected. */
int msm_hip_launch(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, int slot);

/* ---- batch: `batch` independent scalar vectors (batch x n x 32 B, contiguous, device memory) over the resident bases;
 *      out: batch x 96 B.  Internally a software pipeline over the result slots (BASELINE.json config 5: many MSMs over
 *      one shared base). ---- */
int msm_hip_run_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, size_t batch, uint8_t* out_xyz);
/* same with the scalar vectors in host memory (batch x n x 32 B): each vector is copied to the device just ahead of its own
 * MSM, inside the same pipeline */
int msm_hip_run_batch(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz);

/* ---- window-sharded execution (multi-GPU; Pippenger windows are independent).
 *      Computes the window sums S_w for w in [w_begin, w_end) and writes (w_end - w_begin) x 96 B Jacobian
 *      canonical-LE records to `window_sums_dev` (device memory, so that RCCL can gather them in place). ---- */
int msm_hip_run_windows_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end,
                                     void* window_sums_dev);
/* asynchronous form: enqueue into a result slot and return.  window_sums_dev == NULL keeps the sums in the slot and
 * copies them to the host (then msm_hip_finish applies when all 16 windows were run).  Afterwards:
 *   msm_hip_slot_wait_stream  makes a foreign HIP stream (e.g. the one RCCL runs on) wait for the slot on the device,
 *                             without blocking the host;
 *   msm_hip_slot_sync         blocks the host until the slot is complete and returns its error status. */
int msm_hip_launch_windows_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end, int slot,
                                        void* window_sums_dev);
/* several MSMs per launch: `nvec` scalar vectors (contiguous, nvec x n x 32 B) over the resident bases, windows
 * [w_begin, w_end) of each, nvec * (w_end - w_begin) <= MSM_HIP_MAX_LOCAL_WINDOWS; window_sums_dev receives nvec x (w_end - w_begin) x 96 B
 * (vector-major).  One kernel sequence sorts, accumulates and reduces all of them: a rank of a window-sharded run whose
 * own share (2 windows of one MSM at 8 GPUs) cannot fill the GPU processes 8 MSMs' shares at once.
 * Up to MSM_HIP_MAX_LOCAL_WINDOWS = 64 local windows per launch, i.e. also up to 4 WHOLE small MSMs (w_begin = 0, w_end = 16,
 * window_sums_dev = NULL): msm_hip_finish_batch then waits for the slot and writes nvec x 96 B results.  The batch
 * entry points above group small MSMs this way on their own. */
int msm_hip_launch_windows_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int w_begin,
                                              int w_end, int slot, void* window_sums_dev);
int msm_hip_finish_batch(msm_hip_ctx* ctx, int slot, uint8_t* out_xyz);
/* the same for a context whose bases were set with MSM_HIP_BASES_ENDOMORPHISM: HALF-length windows [hw_begin, hw_end) of the 8 that the
 * 127-bit halves k1, k2 of every scalar have (k = k1 + k2 lambda; the MSM runs over the 2n points P_i, phi(P_i)): at 8 GPUs one such window
 * per rank instead of two full-length ones -- the same bucket additions, half the buckets to stitch and reduce.  The 8 sums S_hw combine like
 * any window sums: result = sum_hw 2^(16 hw) S_hw (msm_hip_combine_windows_bn254 with num_windows = 8).  Anchor: the reference runs 16
 * full-length windows, src/cuzk/msm.rs:79-82. */
int msm_hip_launch_half_windows_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int hw_begin,
                                                   int hw_end, int slot, void* window_sums_dev);
/* the same for a context whose bases were set with MSM_HIP_BASES_PRECOMPUTE_WIDE (round 5: the fixed-base tables behind the window-sharded paths; anchors:
 * README.md:70-71 -- the time / space trade-off --, src/cuzk/msm.rs:79-82 -- chunk_size hard-coded --, src/cuzk/msm.rs:411-416 -- the final combine):
 * the ranks share the 2^(C-16) VIRTUAL windows of the one bucket set (msm_hip_wide_config: 8 at 19-bit digits, 16 at 20).  Virtual windows
 * [v_begin, v_end) of every vector: at 8 GPUs and 19 bits ONE bucket set of 2^15 slots and 14 n / 8 entries per rank and MSM, instead of two sets and
 * 2 n entries with 16-bit windows.  `sums_dev` (device memory; NULL: the slot's pinned buffer, for msm_hip_mgpu's host gather) receives
 * nvec x (v_end - v_begin) x 2 records, vector-major: for every virtual window its weighted sum W_hi and its plain total TC_hi.  The gathered
 * pairs of all virtual windows finish as  sum_hi W_hi + 2^15 sum_hi hi TC_hi  (msm_hip_combine_vwindows_batch_curve; pairs_host: nvec x
 * num_vwindows x 2 records, out_xyz: nvec records).  nvec x (v_end - v_begin) <= MSM_HIP_MAX_LOCAL_WINDOWS. */
int msm_hip_launch_vwindows_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int v_begin, int v_end, int slot,
                                         void* sums_dev);
int msm_hip_combine_vwindows_batch_curve(int curve, const uint8_t* pairs_host, int num_vwindows, int nvec, uint8_t* out_xyz);
int msm_hip_slot_wait_stream(msm_hip_ctx* ctx, int slot, void* hip_stream);
int msm_hip_slot_sync(msm_hip_ctx* ctx, int slot);
/* result = sum_w 2^(16 w) * S_w over num_windows records (host memory): src/cuzk/msm.rs:411-416 */
int msm_hip_combine_windows_bn254(const uint8_t* window_sums_host, int num_windows, uint8_t out_xyz[96]);
/* the same for `nvec` MSMs at once (window_sums_host: nvec x num_windows x 96 B, out_xyz: nvec x 96 B): the independent Horner chains
 * (~47 us each) run side by side on a small pool of host threads the library keeps (MSM_HIP_COMBINE_THREADS=1: serially) */
int msm_hip_combine_windows_batch_curve(int curve, const uint8_t* window_sums_host, int num_windows, int nvec, uint8_t* out_xyz);

/* ---- multi-GPU in ONE host process (BASELINE.json north star: "independent Pippenger windows shard across the 8 GPUs of one
 *      node with a final RCCL gather/reduce of partial sums over xGMI"; the reference is single-device, src/cuzk/msm.rs:88-94).
 *      One engine context per listed device.  msm_hip_mgpu_run: device d computes the window sums of window range d
 *      (msm_hip_window_range(d, n_devices, 16, ...); bases replicated, every device receives all scalars), the sums are gathered
 *      -- ncclAllGather over RCCL (librccl is loaded at run time; not a link dependency) or through the slots' pinned result
 *      buffers -- and the host window combine (src/cuzk/msm.rs:411-416) runs once.  msm_hip_mgpu_run_batch deals whole
 *      MSMs out contiguously (BASELINE config 5: many MSMs over one shared base): no exchange at all.
 *      Device ids may repeat with MSM_HIP_MGPU_GATHER_HOST (several contexts on one GPU: rehearsal on a one-GPU box). ---- */
typedef struct msm_hip_mgpu msm_hip_mgpu;
#define MSM_HIP_MGPU_GATHER_AUTO 0u /* RCCL if more than one distinct device and librccl loads, else the pinned-buffer gather */
#define MSM_HIP_MGPU_GATHER_HOST 1u /* every device's sums leave through its slot's pinned buffer                        */
#define MSM_HIP_MGPU_GATHER_RCCL 2u /* ncclAllGather; creation fails if RCCL cannot be initialised on these devices       */
int msm_hip_mgpu_create(msm_hip_mgpu** out, const int* device_ids, int n_devices, uint32_t gather_flags);
int msm_hip_mgpu_create_curve(msm_hip_mgpu** out, const int* device_ids, int n_devices, uint32_t gather_flags, int curve); /* MSM_HIP_CURVE_* */
void msm_hip_mgpu_destroy(msm_hip_mgpu* m);
int msm_hip_mgpu_device_count(const msm_hip_mgpu* m);
int msm_hip_mgpu_uses_rccl(const msm_hip_mgpu* m);
int msm_hip_mgpu_set_bases(msm_hip_mgpu* m, const uint8_t* xy_host, size_t n, uint32_t flags);
int msm_hip_mgpu_run(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]);
/* The throughput form of the same (what a Rust caller that holds many (points, scalars) jobs drives, src/lib.rs:76-82; the reference creates
 * its device per call, src/cuzk/msm.rs:88-94): `nvec` scalar vectors share ONE launch -- device d runs its window range of every vector in one
 * kernel sequence (a single MSM's share, 2 of 16 windows at 8 GPUs, cannot fill a GPU) into result slot `slot` (0 .. MSM_HIP_NUM_SLOTS-1) of its
 * context, and ONE all-gather per launch follows in stream order.  `launch` returns at once: every device has a persistent host thread that
 * issues its calls, so several slots can be in flight.  `finish` waits for the slot and writes nvec x 96 B results (one host window combine
 * per MSM, side by side on the host pool).  nvec x (windows per device) <= MSM_HIP_MAX_LOCAL_WINDOWS; msm_hip_mgpu_group_size() is the nvec
 * that fills a device (8 at 8 GPUs).  With bases set with MSM_HIP_BASES_ENDOMORPHISM the shares are the 8 half-length windows.
 *   launch_batch         : scalars in host memory (nvec x n x 32 B); every device uploads all of them (PCIe-bound: the latency form).  The buffer
 *                    must stay untouched until finish.
 *   launch_batch_device  : scalars_dev[d] = the same nvec x n x 32 B already resident on device d (complete before the call; alive until finish).
 * Bases (msm_hip_mgpu_set_bases): replicated; the flags decide what the devices share -- MSM_HIP_BASES_PLAIN or 0: the reference's 16 windows over
 * the n points (flags = 0 is resolved to PLAIN here, not to a context's own default: the shares of full-length windows read only the plain records);
 * MSM_HIP_BASES_ENDOMORPHISM: the 8 half-length windows; MSM_HIP_BASES_PRECOMPUTE_WIDE: the virtual windows of the wide tables (digit width by
 * msm_hip_mgpu_set_wide_bits, default 19: 8 virtual windows, 14 bucket additions per point, one bucket set per device at 8 GPUs).
 * A failing launch is reported by finish, which always leaves the slot free.  RCCL gather: the devices' calls of a launch's all-gather
 * are issued in lock-step -- a device whose own launch failed (slot busy, out of memory, a HIP error) still enters the collective, with a
 * zeroed block -- so a failure on one device can neither hang the others' collectives nor shift the pairing of later launches; finish
 * returns that device's error within the time of a normal launch and the next launch runs normally.  Only a collective that could not be
 * ISSUED on some device leaves the object unusable (every later call returns MSM_HIP_ERR_HIP; destroy aborts the communicator). */
int msm_hip_mgpu_launch_batch(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, int nvec, int slot);
int msm_hip_mgpu_launch_batch_device(msm_hip_mgpu* m, const void* const* scalars_dev, size_t n, int nvec, int slot);
int msm_hip_mgpu_finish_batch(msm_hip_mgpu* m, int slot, uint8_t* out_xyz);
int msm_hip_mgpu_group_size(const msm_hip_mgpu* m);
/* digit width (16 .. 20; 0: 19) of the wide tables the next msm_hip_mgpu_set_bases(..., MSM_HIP_BASES_PRECOMPUTE_WIDE) builds on every device */
int msm_hip_mgpu_set_wide_bits(msm_hip_mgpu* m, int bits);
int msm_hip_mgpu_run_batch(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz);
/* contiguous balanced partition of [0, num) over `world` ranks (the first num % world ranks take one more): the window ranges of
 * msm_hip_mgpu_run and the MSM ranges of msm_hip_mgpu_run_batch; host-only */
int msm_hip_window_range(int rank, int world, int num, int* begin, int* end);

/* host-only helper (≙ Curve::to_affine as used by tests/cuzk.rs:88-94): 96 B Jacobian -> 64 B canonical affine x || y.
 * Returns 1 when the point is the identity (out zeroed), 0 otherwise, negative on error. */
int msm_hip_g1_to_affine_bn254(const uint8_t xyz[96], uint8_t out_xy[64]);

/* ---- one-shot: context, set bases, run (≙ compute_msm as the reference calls it, src/cuzk/msm.rs:75-94), on the caller's current
 *      device.  The library keeps the context it used (streams, device pools) for the next one-shot call on that device instead of
 *      creating and destroying one per call as the reference does with its wgpu device (creation, first-use allocations and the frees
 *      cost more than a 2^20 MSM); calls are serialised by a process-wide mutex, results do not depend on it.
 *      msm_hip_oneshot_release() frees the kept contexts (call it before unloading the library or to return the device memory);
 *      MSM_HIP_ONESHOT_KEEP=0 in the environment restores create / destroy per call.
 *      Round 5: the scalars are uploaded first and sorted while the points arrive in chunks, and from 2^19 points on the call runs as TWO sub-MSMs over
 *      the two halves of the points (a context each, uploads one behind the other): the first half accumulates while the second is still arriving, and
 *      only the second half's accumulation follows the upload; the two results are added on the host.  The result is the same group element. ---- */
int msm_hip_msm_bn254_g1(const uint8_t* xy_host, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]);
/* the same on any curve of this library (MSM_HIP_CURVE_*; record sizes as that curve's): msm_hip_msm_bn254_g1 is curve 0 */
int msm_hip_msm_curve(int curve, const uint8_t* xy_host, const uint8_t* scalars_host, size_t n, uint8_t* out_xyz);
void msm_hip_oneshot_release(void);

/* ---- synthetic inputs generated in HBM (≙ sample_scalars / sample_points, src/lib.rs:20-42, seeded):
 *      scalars uniform in [0, r) by rejection; points by try-and-increment on x (Curve::random does the same).
 *      Output is the wire format above, written to device memory. ---- */
int msm_hip_sample_scalars_device(msm_hip_ctx* ctx, uint64_t seed, size_t n, void* scalars_dev);
int msm_hip_sample_points_device(msm_hip_ctx* ctx, uint64_t seed, size_t n, void* xy_dev);

/* ---- measurement: per-stage device time of the last finished run, from HIP events on the context's own stream.
 *      ms[0] recode + coarse histogram, ms[1] coarse scan, ms[2] coarse scatter, ms[3] fine sort,
 *      ms[4] SMVP accumulate (k_smvp_chunks), ms[5] SMVP stitch, ms[6] bucket reduce,
 *      ms[7] whole device pipeline, ms[8] host finalisation.  Returns the number of entries written. ---- */
int msm_hip_last_stage_ms(msm_hip_ctx* ctx, float* ms, int cap);
/* which stage boundaries get HIP events (each costs a few microseconds of queue time between kernels):
 * 0 none, 1 only around the SMVP accumulate kernel (ms[4]; everything else reads 0), 2 every stage (default) */
int msm_hip_set_stage_timing(msm_hip_ctx* ctx, int level);
/* the context's main stream as a hipStream_t */
void* msm_hip_stream(msm_hip_ctx* ctx);

/* ==== TEST HOOKS ========================================================================================================================
 * Everything between here and the matching #endif exists for the parity tests and the profiling scripts only -- stage read-back, single-operation
 * kernels, the deterministic transpose, fault injection.  A binding of the product API does not need them: they are declared only with
 * MSM_HIP_TEST_HOOKS defined (the symbols are exported either way). */
#ifdef MSM_HIP_TEST_HOOKS
/* ---- stage-level read-back for parity tests (≙ read_from_gpu_test, src/cuzk/gpu.rs:137-171).  Each copies the
 *      buffer left by the last run to host memory.  Layouts:
 *      digits    : u16[num_windows_run][n]    code = sign << 15 | (|d| & 0x7fff); 0 = digit 0 (no entry);
 *                  0x8000 = digit -2^15 (bucket slot 0)        cf. decompose_scalars.template.wgsl:93-112
 *                  (endomorphism launches: 2n inputs per window, input 2 j = first half of scalar j, 2 j + 1 = its second half)
 *      (shapes for 16-bit windows; with b-bit windows: 2^(b-1) buckets per window, col_ptr rows of 2^(b-1) + 1 entries)
 *      col_ptr   : u32[num_windows_run][32769] start of bucket slot k in val_idxs   cf. transpose.template.wgsl:58-61
 *      val_idxs  : u32[num_windows_run][n]    point index | sign << 31, grouped by slot (order within a slot is
 *                  unspecified)                                 cf. transpose.template.wgsl:66-73
 *      buckets   : [num_windows_run][32768] x 96 B Jacobian canonical LE, slot k as smvp.template.wgsl:94
 *      windows   : [num_windows_run] x 96 B Jacobian canonical LE (a single-MSM launch whose sums the host combines leaves the
 *                  bucket reduce's 16 bit-plane sums per window on the device, k_bpr_planes: the read-back finishes them)
 * ---- */
/* digit-code planes are only materialised for read-back when enabled here (the sort recomputes digits on the fly) */
int msm_hip_set_debug(msm_hip_ctx* ctx, int keep_digit_planes);
/* from `n` points on, the fine sort gets the sub-range histograms of huge coarse bins from a separate pass (default 32769:
 * whenever such a bin can exist); tests raise it to drive the fallback in which every sharer of a bin histograms it itself */
int msm_hip_set_fine_hist_min_n(msm_hip_ctx* ctx, size_t n);
int msm_hip_read_digits(msm_hip_ctx* ctx, uint16_t* out, size_t cap_elems);
int msm_hip_read_col_ptr(msm_hip_ctx* ctx, uint32_t* out, size_t cap_elems);
int msm_hip_read_val_idxs(msm_hip_ctx* ctx, uint32_t* out, size_t cap_elems);
int msm_hip_read_buckets(msm_hip_ctx* ctx, uint8_t* out, size_t cap_bytes);
int msm_hip_read_window_sums(msm_hip_ctx* ctx, uint8_t* out, size_t cap_bytes);

/* ---- device op hooks for parity tests (≙ the single-op shaders src/cuzk/wgsl/test/test_field.wgsl:13-62 and
 *      test_point.wgsl:18-88 driven by tests/field.rs:68, tests/point.rs:71).  Host buffers, canonical LE.
 *      fq op: 0 add, 1 sub, 2 mul, 3 sqr, 4 neg ; a, b, out: n x 32 B
 *             5 mul, 6 sqr, 7 a*b + b*a, 8 (a+b)*2a, 9 (a+b)^2 through the SMVP's inline-assembly multipliers (fq29_asm.h)
 *      g1 op: 0 add, 1 double(a), 2 a + affine(b: n x 64 B) ; a, b, out: n x 96 B Jacobian
 *             3 a + b - b + b, 4 a - b - b through the SMVP's signed-state mixed addition (g1_madd_w, csrc/g1.h)
 *      g1_mul_u32: out[i] = k[i] * a[i] (≙ double_and_add) ---- */
int msm_hip_test_fq_op(msm_hip_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
int msm_hip_test_g1_op(msm_hip_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
int msm_hip_test_g1_mul_u32(msm_hip_ctx* ctx, const uint8_t* a, const uint32_t* k, uint8_t* out, size_t n);

/* test hook: the next `launches` window-sharded launches fail on device index `device_index` (MSM_HIP_ERR_HIP, before anything is queued
 * there) -- the rehearsal of one failing GPU.  Armed only through this call (round 5: no environment variable can make a deployment's launches fail). */
int msm_hip_mgpu_inject_fault(msm_hip_mgpu* m, int device_index, int launches);

/* test hook: the part policy of the one-shot call (msm_hip_msm_curve: `parts` sub-MSMs over ranges of the points when n >= min_points; 0, 0 restores
 * the default of 2 parts from 2^19 points on) -- so that a test reaches the multi-part path with inputs the oracle finishes in seconds. */
int msm_hip_test_oneshot_parts(int parts, size_t min_points);
#endif /* MSM_HIP_TEST_HOOKS */

const char* msm_hip_strerror(int code);
int msm_hip_last_hip_error(msm_hip_ctx* ctx);
/* ABI version; bumped on any signature change or addition (7 = round 5: curve-neutral names with the `_bn254` aliases kept, the virtual-window
 * launches and their pair combine, msm_hip_msm_curve, msm_hip_mgpu_set_wide_bits) */
int msm_hip_abi_version(void);

/* ---- aliases: the names of rounds 1 - 4 (the reference instantiates its generic functions with halo2curves::bn256 only, src/lib.rs:91,154). Each is
 *      the curve-neutral entry point of the same name without `_bn254`, on whatever curve the context was created for. ---- */
int msm_hip_set_bases_bn254(msm_hip_ctx* ctx, const uint8_t* xy_host, size_t n, uint32_t flags);
int msm_hip_set_bases_device_bn254(msm_hip_ctx* ctx, const void* xy_dev, size_t n, uint32_t flags);
int msm_hip_run_bn254(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]);
int msm_hip_run_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, uint8_t out_xyz[96]);
int msm_hip_launch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int slot);
int msm_hip_finish_bn254(msm_hip_ctx* ctx, int slot, uint8_t out_xyz[96]);
int msm_hip_launch_bn254(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, int slot);
int msm_hip_run_batch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, size_t batch, uint8_t* out_xyz);
int msm_hip_run_batch_bn254(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz);
int msm_hip_run_windows_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end, void* window_sums_dev);
int msm_hip_launch_windows_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end, int slot, void* window_sums_dev);
int msm_hip_launch_windows_batch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int w_begin, int w_end, int slot,
                                              void* window_sums_dev);
int msm_hip_finish_batch_bn254(msm_hip_ctx* ctx, int slot, uint8_t* out_xyz);
int msm_hip_launch_half_windows_batch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int hw_begin, int hw_end, int slot,
                                                   void* window_sums_dev);
int msm_hip_mgpu_set_bases_bn254(msm_hip_mgpu* m, const uint8_t* xy_host, size_t n, uint32_t flags);
int msm_hip_mgpu_run_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]);
int msm_hip_mgpu_launch_batch_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, int nvec, int slot);
int msm_hip_mgpu_launch_batch_device_bn254(msm_hip_mgpu* m, const void* const* scalars_dev, size_t n, int nvec, int slot);
int msm_hip_mgpu_finish_batch_bn254(msm_hip_mgpu* m, int slot, uint8_t* out_xyz);
int msm_hip_mgpu_run_batch_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz);

#ifdef __cplusplus
}
#endif
#endif /* MSM_HIP_H */
