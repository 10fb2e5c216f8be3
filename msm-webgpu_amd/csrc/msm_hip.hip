// libmsm_hip.so -- host side of the MI355X BN254 MSM engine behind the C ABI of include/msm_hip.h.
//
// Replaces, for the hot path only, the reference's orchestrator compute_msm (src/cuzk/msm.rs:75-417) and its wgpu
// wrappers (src/cuzk/gpu.rs): one persistent context = three HIP streams, pooled device buffers, bases resident in HBM
// in device Montgomery form; no per-call device creation, shader generation or pipeline compilation
// (cf. src/cuzk/msm.rs:88-94, src/cuzk/shader_manager.rs:74-100).
//
// Execution model.  Every launch (one MSM, or several scalar vectors sharing one kernel sequence: up to MAXLW local
// windows) runs in one of MSM_HIP_NUM_SLOTS result slots (own bucket, piece, col_ptr and window-sum buffers):
//   stream "main"   : recode + sort + SMVP accumulate                               -> event smvp_done[slot]
//   stream "reduce" : (waits smvp_done) stitch + bucket reduce -> window sums -> D2H -> event done[slot]
// The stitch and the bucket reduce are bound by the depth of dependent group additions and occupy few waves; putting them
// on their own stream lets the sort + SMVP of the NEXT launch (other slot) run meanwhile.  The host window combine of a
// slot (src/cuzk/msm.rs:411-416) runs in the caller's thread inside msm_hip_finish / msm_hip_finish_batch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#define MSM_HIP_TEST_HOOKS 1
#include "../../include/msm_hip.h"
#include "curve_ops.h"
#include "host_pool.h"

// the arithmetic and the kernels of BN254's unit (csrc/curve_select.h); the other curves' units are separate translation units (curve_ops.h)
#define MSM_FIELD_NS bn254
#define MSM_KERNEL_NS msmk
#define MSM_CURVE_CONSTANTS "bn254_constants.h"
#include "curve_unit.h"
#undef MSM_FIELD_NS
#undef MSM_KERNEL_NS
#undef MSM_CURVE_CONSTANTS

using namespace msmk;  // layout constants and the field-independent kernels (recode, sort) are taken from BN254's unit

namespace {
const CurveOps BN254_OPS = MSM_CURVE_OPS(msmk, bn254);
// curve id (MSM_HIP_CURVE_*) -> its table
inline const CurveOps* curve_ops(int curve) {
  switch (curve) {
    case MSM_HIP_CURVE_GRUMPKIN: return msm_hip_curve_ops_grumpkin();
    case MSM_HIP_CURVE_PALLAS: return msm_hip_curve_ops_pallas();
    case MSM_HIP_CURVE_VESTA: return msm_hip_curve_ops_vesta();
    case MSM_HIP_CURVE_BLS12_381: return msm_hip_curve_ops_bls12_381();
    case MSM_HIP_CURVE_BN254_G2: return msm_hip_curve_ops_bn254_g2();
    case MSM_HIP_CURVE_BLS12_381_G2: return msm_hip_curve_ops_bls12_381_g2();
    default: return &BN254_OPS;
  }
}

// What a launch's local windows are made of
enum LaunchMode {
  MODE_PLAIN = 0,   // windows [w_begin, w_end) of 254-bit scalars over the n bases
  MODE_TABLES = 1,  // fixed-base tables: all windows of a vector feed one bucket set (MSM_HIP_BASES_PRECOMPUTE)
  MODE_HALVES = 2,  // endomorphism: 127-bit halves k1, k2 over the 2n points P_i, phi(P_i) (MSM_HIP_BASES_ENDOMORPHISM, csrc/glv.h)
  MODE_WIDE = 3,    // wide fixed-base tables: ceil(255 / C) digits of C = 16 .. 20 bits per scalar, one bucket set of 2^(C-1) slots run as 2^(C-16) virtual windows of 2^15
                    // (MSM_HIP_BASES_PRECOMPUTE_WIDE; msm_kernels.h: k_count_wide)
};

constexpr int N_MAIN_EVENTS = 7;  // boundaries of the 6 timed stages on the main stream
constexpr uint32_t MAX_TILES = 1024;
constexpr size_t MAX_JB = 288;  // the largest Jacobian record of any curve (BLS12-381 G2: 3 x 96 B; BN254 G2: 3 x 64 B; BLS12-381 G1: 3 x 48 B; the 254 / 255-bit G1 curves: 96 B)
constexpr size_t WSUM_BYTES = (size_t)24 * PLANES_PER_WINDOW * MAX_JB;  // MAXLW window sums or, for one host-combined MSM, the bit-plane sums (k_bpr_planes) of its <= 22 windows
static_assert(WSUM_BYTES >= (size_t)2 * MAXLW * MAX_JB, "window-sum buffer (pairs of records for shares of the wide tables' virtual windows)");
constexpr int NSLOT = MSM_HIP_NUM_SLOTS;  // result slots
constexpr int NREDUCE = 2;  // reduce streams (slot k uses stream k % NREDUCE): two bucket reduces may be in flight when the
                            // main-stream work of one MSM is shorter than its bucket reduce (few windows per GPU).  The context
                            // then owns 3 streams; callers that add their own (copies, RCCL) should raise GPU_MAX_HW_QUEUES
                            // above ROCm's default of 4 so that streams do not share hardware queues (bench.py does).

struct Slot {
  uint8_t* h_wsums = nullptr;      // pinned: MAXLW x 96 B window sums + 4 B error word
  uint8_t* d_wsums = nullptr;      // device: MAXLW x 96 B window sums + 4 B error word
  uint32_t* d_buckets = nullptr;   // [cap_lw][32768] XYZZ records
  size_t cap_buckets = 0;          // bucket records d_buckets holds (local windows x slots per window of the largest launch seen)
  int wbits = WBITS;               // window bits of the launch in this slot
  uint32_t* d_partials = nullptr;  // bucket-reduce scratch: [W][256] row sums, [W][256] column sums, [W][3] parts (XYZZ)
  uint32_t* d_col_ptr = nullptr;   // [W][32769] start of every bucket slot's run in the sorted entry list
  uint32_t* d_heads = nullptr;     // [W][chunks] XYZZ records: SMVP pieces of runs that cross chunk boundaries
  uint32_t* d_tails = nullptr;     // [W][chunks] XYZZ records
  uint32_t* d_big_queue = nullptr;    // buckets with many pieces (skewed scalars): [0] = count, items, arrival counters, scratch records (BIGQ_*)
  hipEvent_t ev[N_MAIN_EVENTS] = {};
  hipEvent_t red0 = nullptr, red1 = nullptr;  // bucket reduce begin / end on the reduce stream
  hipEvent_t smvp_done = nullptr;             // main -> reduce hand-off
  hipEvent_t done = nullptr;                  // everything of this slot finished (recorded on the reduce stream)
  hipEvent_t staged = nullptr;                // host scalars of this slot have landed in d_host_scalars (copy stream)
  uint32_t* d_host_scalars = nullptr;         // staging of msm_hip_launch's host scalars (this slot's own: no launch of
  size_t cap_host_scalars = 0;                // another slot can still be reading it), in scalars; allocated on first use
  size_t cap_recs = 0;                        // capacity (records) of d_heads / d_tails
  bool ready = false;                         // small buffers + events exist (slots are set up on first use)
  bool timed = false, pending = false, to_host = false;
  bool merged = false;                        // fixed-base launch: one bucket set (one window sum) per scalar vector
  bool halves = false;                        // endomorphism launch: the windows are those of 127-bit halves
  int wide_bits = 0;                          // wide fixed-base launch (its digit width): h_wsums holds the bit-plane sums of the virtual windows (combine_wide)
  bool parts = false;                         // h_wsums holds the bit-plane sums of every window (k_bpr_planes): the host finishes the window sums
  bool pairs = false;                         // a share of the wide tables' virtual windows: the launch leaves (window sum, plain total) record pairs
  int timing_level = 0;
  int w_begin = 0, w_count = 0, nvec = 1;  // windows [w_begin, w_begin + w_count) of nvec scalar vectors
  size_t n = 0;
  size_t base_off = 0;  // first base of this launch (records): point i of the launch is base base_off + i
};

}  // namespace

namespace {
// Part policy of the upload-bound call shapes (msm_hip_msm_curve: bases + scalars from the host; msm_hip_run: scalars from the host): 0 = the default
// (MSM_HIP_ONESHOT_PARTS, else 2 parts from 2^19 points on); set by the test hook msm_hip_test_oneshot_parts
std::atomic<int> g_oneshot_parts{0};
std::atomic<size_t> g_oneshot_parts_min_n{0};
// `two_from`, `three_from`: log2 of the point counts from which the call shape runs 2 / 3 parts (measured: profiles/r05_oneshot.txt)
inline int upload_parts(size_t n, int cap, int two_from, int three_from) {
  static const int env_parts = [] { const char* e = getenv("MSM_HIP_ONESHOT_PARTS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 4 ? v : 0; }();
  const int hook = g_oneshot_parts.load();
  const size_t hook_n = g_oneshot_parts_min_n.load();
  int want = hook ? hook : env_parts;
  if (!want) want = n >= ((size_t)1 << three_from) ? 3 : 2;
  if (want > cap) want = cap;
  const size_t min_n = hook_n ? hook_n : ((size_t)1 << two_from);
  return n >= min_n && n >= (size_t)want ? want : 1;
}
}  // namespace

struct msm_hip_ctx {
  int device = 0;
  int curve = MSM_HIP_CURVE_BN254_G1;
  const CurveOps* ops = nullptr;
  size_t cb = 32, pb = 64, jb = 96;     // bytes of a coordinate, an affine point and a Jacobian record on this curve's wire (48 / 96 / 144: BLS12-381)
  hipStream_t stream = nullptr;         // main
  hipStream_t reduce_stream[NREDUCE] = {};  // bucket reduce + result copies
  int last_hip_error = 0;

  uint32_t* d_bases = nullptr;  // n_bases x 16 words
  size_t launch_base_off = 0;   // consumed by the next launch: it runs over the bases [off, off + n) (internal: the parts of msm_hip_run)
  size_t n_bases = 0, cap_bases = 0;  // points per table; capacity in point records (16 x n_bases with fixed-base tables)
  bool precomputed = false;           // d_bases holds the 16 tables 2^(16 w) P_i (MSM_HIP_BASES_PRECOMPUTE)
  int wide_bits_choice = 0;           // msm_hip_set_wide_bits: the digit width the next wide base set gets (0: by the number of bases)
  int wide_bits = 0;                  // != 0: d_bases holds the wide tables 2^(C w) P_i for C-bit digits (MSM_HIP_BASES_PRECOMPUTE_WIDE; pick_wide_bits)
  bool endo = false;                  // d_bases holds phi(P_i) behind the n bases (MSM_HIP_BASES_ENDOMORPHISM)
  uint32_t* d_halves = nullptr;       // the split scalars of one launch (main stream only): [vector][2n] x 4 words
  size_t cap_halves = 0;              // in scalars

  hipStream_t copy_stream = nullptr;    // H2D of host scalars (created on first use by msm_hip_launch)
  hipEvent_t input_ready = nullptr;     // a caller's producer stream -> main stream (msm_hip_wait_stream)
  hipEvent_t bases_ready = nullptr;     // the one-shot entry point: the last chunk of the bases has been converted (conversion stream -> main stream)
  hipEvent_t chunk_landed[8] = {};      // ... and chunk k of the wire bytes has landed (copy stream -> conversion stream)
  size_t cap_entries = 0;  // capacity of the entry arrays (tmp_val, tmp_fine, val): local windows x per-window stride
  size_t last_stride = 0;  // per-window stride of the last launch (n rounded up to a multiple of 4)
  size_t cap_chunk_slot = 0;  // capacity (records) of d_chunk_slot
  uint8_t* d_batch_stage = nullptr;   // staging ring (NSLOT vectors) of msm_hip_run_batch, allocated on first use
  size_t cap_batch_stage = 0;
  uint16_t* d_digits = nullptr;  // digit-code planes [local window][n]: debug read-back, or the input of the second sort pass (k_scatter_planes)
  uint64_t* d_negbits = nullptr; // with the planes of a launch: one sign bit per input of every vector
  size_t cap_planes = 0;         // capacity (u16 entries) of d_digits; d_negbits holds cap_planes / 64 + 2 MAXLW words
  bool debug = false;
  bool sync_call = false;  // set by the synchronous entry points (run = launch + finish at once) around their launch: nothing will be pipelined behind it
  int timing_level = 2;  // 0: no stage events, 1: only around the SMVP kernel, 2: every stage boundary
  uint32_t* d_counts = nullptr;      // [W][tiles][128]
  uint32_t* d_bin_total = nullptr;   // [W][128]
  uint32_t* d_coarse_ptr = nullptr;  // [W][129]
  uint32_t* d_tmp_val = nullptr;     // [W][stride] coarse-bin order
  uint8_t* d_tmp_fine = nullptr;     // [W][stride]
  uint32_t* d_val = nullptr;         // [W][stride] slot order
  uint32_t* d_chunk_slot = nullptr;  // [W][chunks] bucket slot of every SMVP chunk's first entry
  uint32_t* d_list_len = nullptr;    // shares of the wide tables' virtual windows: [W][sub-tiles] lengths of the first pass's compact entry lists (k_count_wide_list)
  size_t cap_list_len = 0;
  uint32_t* d_scalar_conv = nullptr;  // canonical copies of scalars handed over in R = 2^256 Montgomery form (one launch's worth)
  size_t cap_scalar_conv = 0;         // in scalars
  uint32_t scalar_format = 0;         // MSM_HIP_SCALARS_CANONICAL / MSM_HIP_SCALARS_MONT256
  int window_bits = 0;                // 0: chosen from n for whole-MSM launches (pick_window_bits); else 12 / 14 / 16
  uint32_t* d_part_hist = nullptr;  // [MAXLW][128][FINE_SPLIT][256] sub-range histograms of huge coarse bins (k_fine_hist), on first use
  size_t fine_hist_min_n = FINE_BIG + 1;  // any n that can produce a coarse bin beyond FINE_BIG: run k_fine_hist (3 us when none does)
  int skew_credit = 0;  // launches left for which k_fine_hist runs although uniform scalars could not fill a bin: set when a launch met a huge bin (wait_slot)
  uint32_t* d_err = nullptr;
  uint8_t* d_stage = nullptr;  // staging for host byte inputs of set_bases / test hooks
  size_t cap_stage = 0;

  Slot slot[NSLOT];
  // description of the last launched run (for the stage read-back hooks)
  size_t last_n = 0;
  int last_w_count = 0, last_slot = 0, last_wbits = WBITS;
  bool last_has_digits = false;
  float stage_ms[10] = {};
};

namespace {

#define HIP_TRY(ctx, expr)                                                             \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      if (ctx) (ctx)->last_hip_error = (int)e_;                                        \
      return e_ == hipErrorOutOfMemory ? MSM_HIP_ERR_OUT_OF_MEMORY : MSM_HIP_ERR_HIP; \
    }                                                                                  \
  } while (0)

template <typename T>
int dev_alloc(msm_hip_ctx* ctx, T*& p, size_t count) {
  if (p) {
    (void)hipFree(p);
    p = nullptr;
  }
  HIP_TRY(ctx, hipMalloc((void**)&p, count * sizeof(T)));
  return MSM_HIP_OK;
}

int ensure_stage(msm_hip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->cap_stage) return MSM_HIP_OK;
  int rc = dev_alloc(ctx, ctx->d_stage, bytes);
  if (rc) {
    ctx->cap_stage = 0;
    return rc;
  }
  ctx->cap_stage = bytes;
  return MSM_HIP_OK;
}

// entries per SMVP lane: about SMVP_TARGET_LANES lanes over all windows of the run, within the kernel's limits
inline size_t target_lanes() {  // MSM_HIP_TARGET_LANES overrides the default for tuning experiments
  static const size_t v = [] {
    const char* e = getenv("MSM_HIP_TARGET_LANES");
    const long x = e ? atol(e) : 0;
    return x >= 1024 ? (size_t)x : (size_t)SMVP_TARGET_LANES;
  }();
  return v;
}
// compute units of the current device (the devices of a node are alike: asked once)
inline size_t num_cus() {
  static const size_t v = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) return (size_t)cus;
    return (size_t)256;
  }();
  return v;
}
// The SMVP is bound by the multiplier, so a SIMD needs time proportional to (waves it is given) x (chunk length) however many of them are
// resident at once, and a workgroup puts one wave on each SIMD of its CU: the kernel takes  ceil(workgroups / CUs) x chunk length  (round 3
// sweep of the length at 2^20, profiles/r03_chunk_len_sweep.txt: a sawtooth of 6.5 % that this expression reproduces within 1 - 2 %).  Around the
// length that gives about SMVP_TARGET_LANES lanes (+- 25 %: the stitch's work follows the lane count) a length whose product is more than 1 % lower
// than the plain quotient's replaces it (2^20 with 8 or 16 windows: 32, exact, instead of 29: 1.5 % per MSM).  (What the shares of an 8-rank run gained this round -- 3.5 % per MSM at 7 x 2 windows per launch -- came
// from the quotient itself, 25, no longer being rounded up to a multiple of 4, 28.)  MSM_HIP_CHUNK_SEARCH=0: the plain quotient.
inline uint32_t chunk_len_for(size_t n, int w_count) {
  static const bool search = [] { const char* e = getenv("MSM_HIP_CHUNK_SEARCH"); return !e || atoi(e) != 0; }();
  size_t base = (n * (size_t)w_count + target_lanes() - 1) / target_lanes();
  if (base < (size_t)SMVP_CHUNK_MIN) base = SMVP_CHUNK_MIN;
  if (base > (size_t)SMVP_CHUNK_MAX) base = SMVP_CHUNK_MAX;
  if (!search || base <= (size_t)SMVP_CHUNK_MIN) return (uint32_t)base;
  const size_t cus = num_cus();
  auto cost = [&](size_t c) {
    const size_t groups = (size_t)w_count * (((n + c - 1) / c + 255) / 256);
    return (groups + cus - 1) / cus * c;
  };
  size_t lo = base - base / 4, hi = base + base / 4 + 1;
  if (lo < (size_t)SMVP_CHUNK_MIN) lo = SMVP_CHUNK_MIN;
  if (hi > (size_t)SMVP_CHUNK_MAX) hi = SMVP_CHUNK_MAX;
  // (only a length that is more than 1 % better than the plain quotient replaces it: the expression is a model, and a different lane count
  //  moves work between the SMVP and the stitch -- at 2^24, 512 instead of 456 is 0.2 % better by the model and 1.5 % slower measured)
  const size_t base_cost = cost(base);
  size_t best = base, best_cost = base_cost - base_cost / 100;
  for (size_t c = lo; c <= hi; c++) {
    const size_t k = cost(c);
    const size_t d = c > base ? c - base : base - c, bd = best > base ? best - base : base - best;
    if (k < best_cost || (best != base && k == best_cost && d < bd)) {
      best = c;
      best_cost = k;
    }
  }
  return (uint32_t)best;
}
inline uint32_t chunks_for(size_t n, uint32_t chunk_len) { return (uint32_t)((n + chunk_len - 1) / chunk_len); }
// head/tail piece records needed for any run over at most n points: w_count * chunks is largest at w_count = NWIN
inline size_t piece_records_for(size_t n) {
  size_t worst = 0;
  for (int w = 1; w <= NWIN; w++) {
    const size_t r = (size_t)w * chunks_for(n, chunk_len_for(n, w));
    if (r > worst) worst = r;
  }
  return worst;
}

inline size_t stride_for(size_t n) { return (n + 3) & ~(size_t)3; }

// The top digit of the wide tables' recode (msm_kernels.h: wide_digit), from the scalar field's modulus r (its top 64 bits, r >> 192):
// its largest value over the scalars below r -- (r - 1 + the recode's bias below the digit) >> P, P = the digit's position -- decides the shift
// (the largest that keeps the shifted digit within 2^(C-1)), and r / 2^P, the range of a uniform scalar's top digit, how many virtual windows
// the shifted digit spreads over.  A top digit that does not fit after all is rejected by the kernel, never mis-added.
inline uint64_t scalar_modulus_top64(int curve) {
  switch (curve) {
    case MSM_HIP_CURVE_BLS12_381:
    case MSM_HIP_CURVE_BLS12_381_G2: return 0x73eda753299d7d48ull;
    case MSM_HIP_CURVE_PALLAS:
    case MSM_HIP_CURVE_VESTA: return 0x4000000000000000ull;  // both moduli: 2^254 + (a 126-bit number)
    default: return 0x30644e72e131a029ull;                   // BN254's r, and its p (Grumpkin's scalar field): the same top 64 bits
  }
}
inline int wide_top_pos(int bits) { return bits * (wide_tables_of(bits) - 1); }  // bit position of the top digit: 240 / 238 / 252 / 247 / 240 at 16 .. 20 bits (>= 192)
inline uint32_t wide_top_max(int curve, int bits) {
  const uint64_t top = scalar_modulus_top64(curve);
  const int fb = wide_top_pos(bits) - 192;                   // fraction bits of `top` below the digit
  const uint64_t frac = top << (64 - fb);                    // (r mod 2^P) / 2^P as a 64-bit fraction, truncated
  // the bias below the digit / 2^P = 1/2 + 2^(-C-1) + 2^(-2C-1) + ... as a 64-bit fraction: every term of the series that has a bit there, + 2 units
  // for its tail and for the truncation of `frac` -- an UPPER bound (round 4 stopped after two terms + 2^20, which the third term, 2^(63-2C),
  // exceeds: right for the five moduli of this library, not a bound), so a width is at worst refused for a modulus on a carry boundary, never
  // accepted wrongly; tests/test_abi.py compares with the exact big-integer value for every curve and width
  uint64_t bias = 2;
  for (int sh = 63; sh >= 0; sh -= bits) bias += 1ull << sh;
  return (uint32_t)(top >> fb) + (frac + bias < frac ? 1u : 0u);               // + the carry into the digit
}
// Round 5: with INTERLEAVED virtual windows (msm_kernels.h: wide_key) the narrow top digit spreads over the windows by itself and is used as it
// is -- no shift.  (Round 4 shifted it by the largest amount that kept it within 2^(C-1), to spread it over contiguous magnitude ranges;
// MSM_HIP_WIDE_TOP_SHIFT still forces a shift for A/B runs: the kernels and the tables' last step honour it.)
inline int wide_top_shift(int curve, int bits) {
  static const int forced = [] { const char* e = getenv("MSM_HIP_WIDE_TOP_SHIFT"); return e ? atoi(e) : -1; }();  // tuning aid
  if (forced < 0) return 0;
  const uint32_t dmax = wide_top_max(curve, bits), half = 1u << (bits - 1);
  int s = 0;
  while (s + 1 < bits && s < forced && ((uint64_t)dmax << (s + 1)) <= half) s++;
  return s;
}
// Can the curve's scalars be recoded into C-bit digits at all?  The top digit is never negative and becomes a bucket magnitude, so it must not
// pass 2^(C-1): with 17-bit digits (15 x 17 = 255 bits) that holds for BN254, Grumpkin and -- just: their moduli are 2^254 + a 126-bit number, the
// top digit of their largest scalars is exactly 2^16 -- Pallas and Vesta, not for BLS12-381's modulus of 1.8 x 2^254.
inline bool wide_bits_fit(int curve, int bits) { return wide_top_max(curve, bits) <= (1u << (bits - 1)); }
// Digit width of the wide tables for a base set of n points (profiles/r04_wide_tables.txt, same-box A/Bs against the endomorphism mode).  What an
// MSM costs in the pipeline is sort + SMVP + the stitch / reduce work that runs beside the next launch, and the last grows with the bucket sets:
// 16 bits (16 additions per point like every other mode, but ONE bucket set: the 16-bit tables' shape behind these kernels' all-digits-at-once
// scatter) wins up to 2^16 points, where grouped launches and latencies are all stitch / reduce; 17 bits (15 additions per point, 2 bucket sets)
// up to 2^20 points (+4 % at 2^20), 20 bits (13 additions, 16 bucket sets) beyond (2^21: 372 - 380 against 366 - 369 MSM/s with 17 bits; +11 % over the
// endomorphism mode at 2^22, +18 % at 2^24), where the additions are
// all that counts.  19 bits (14 additions, 8 bucket sets; the 7-bit top digit makes <= 128
// giant buckets) lies between them at every size and serves the curve 17 bits cannot (BLS12-381); 18 bits (a 2-bit top digit: 3 giant buckets) loses everywhere.
// msm_hip_set_wide_bits / MSM_HIP_WIDE_BITS = 16 .. 20 override.  -1: the chosen width cannot hold the curve's scalars.
inline int pick_wide_bits(const msm_hip_ctx* ctx, size_t n) {
  static const int forced = [] { const char* e = getenv("MSM_HIP_WIDE_BITS"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 20 ? v : 0; }();
  const int asked = ctx->wide_bits_choice ? ctx->wide_bits_choice : forced;  // msm_hip_set_wide_bits, then the environment
  if (asked) return wide_bits_fit(ctx->curve, asked) ? asked : -1;
  const int bits = n <= ((size_t)1 << 16) ? 16 : n <= ((size_t)1 << 20) ? 17 : 20;
  return wide_bits_fit(ctx->curve, bits) ? bits : 19;
}
// SMVP lanes and lengths of a wide fixed-base launch over n points (msm_kernels.h: k_count_wide).  For uniform scalars every virtual window
// receives (T - 1) n / VWIN entries from the T - 1 full digits, and the windows the shifted top digit reaches their share of its n more: the fullest
// window's expected count F sets the device's chunk length (smvp_chunk_len), so the lanes are planned for F (+ 0.4 % + 64 entries: its
// fluctuation is 0.07 % at 2^20) -- planned for the mean, the length the device settles on would be one entry more than the one the host
// searched for, 4 % at 2^20.  Skewed scalars spread any other way: the arrays' per-window stride (`worst`) and the longest chunk the
// device may pick (`host_len`) cover one window that holds everything.
struct WideShape {
  size_t worst;
  uint32_t chunk_len, chunks, host_len;
};
inline WideShape wide_shape(size_t n, int curve, int bits, int lwin) {  // (lwin local windows share the lanes: nvec whole MSMs x VWIN, or nvec shares x their virtual windows)
  const int WIDE_TABLES = wide_tables_of(bits), WIDE_VWIN = wide_vwin_of(bits);
  WideShape w;
  w.worst = n * (size_t)WIDE_TABLES;
  // interleaved virtual windows (msm_kernels.h: wide_key): consecutive magnitudes go to consecutive windows, so every digit -- the narrow top one
  // included -- spreads evenly: each window expects T n / VWIN entries (a forced top shift of s puts the top digit's n entries into every 2^s-th
  // window only)
  const int shift = wide_top_shift(curve, bits);
  const int top_windows = WIDE_VWIN >> (shift < bits - WBITS ? shift : bits - WBITS);
  static const double slack = [] { const char* e = getenv("MSM_HIP_WIDE_SLACK_PCT"); return e ? atof(e) / 100.0 : 0.004; }();  // tuning aid
  const double fullest = (double)n * (WIDE_TABLES - 1) / WIDE_VWIN + (double)n / (top_windows > 0 ? top_windows : 1);
  const size_t typ = (size_t)(fullest * (1.0 + slack)) + 64;
  w.chunk_len = chunk_len_for(typ, lwin);
  w.chunks = chunks_for(typ, w.chunk_len);
  w.host_len = (uint32_t)((w.worst + w.chunks - 1) / w.chunks);
  if (w.host_len < w.chunk_len) w.host_len = w.chunk_len;
  return w;
}

// RAII: every ABI entry point runs on its context's device and leaves the caller's current device as it found it
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = prev == device || hipSetDevice(device) == hipSuccess;
    if (prev == device) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(ctx)                     \
  DeviceGuard device_guard_((ctx)->device); \
  if (!device_guard_.ok) return MSM_HIP_ERR_NO_DEVICE

// first use of a result slot: its events and small buffers (the big ones -- buckets, pieces -- follow in ensure_work).
// Slots are set up lazily so that a context that only ever runs one MSM at a time (the reference's call shape,
// src/cuzk/msm.rs:75-94: create, run once, destroy) allocates one slot's worth of memory, not four.
int setup_slot(msm_hip_ctx* ctx, Slot& s) {
  if (s.ready) return MSM_HIP_OK;
  int rc;
  if (!s.h_wsums) {
    HIP_TRY(ctx, hipHostMalloc((void**)&s.h_wsums, WSUM_BYTES + 4, hipHostMallocDefault));
    memset(s.h_wsums, 0, WSUM_BYTES + 4);
  }
  if ((rc = dev_alloc(ctx, s.d_wsums, WSUM_BYTES + 4))) return rc;
  if ((rc = dev_alloc(ctx, s.d_partials, (size_t)MAXLW * (256 + 256 + PLANES_PER_WINDOW) * ctx->ops->xyzz_words))) return rc;
  if ((rc = dev_alloc(ctx, s.d_col_ptr, (size_t)MAXLW * (HALF + 1)))) return rc;
  if ((rc = dev_alloc(ctx, s.d_big_queue, BIGQ_SCRATCH + (size_t)STITCH_BLOCKS * ctx->ops->rec_words))) return rc;
  // zeroed on the stream that first reads them (the slot's reduce stream; the error word is first written on the main
  // stream, which waits for `done` below) -- not on the null stream, which the non-blocking streams do not order with
  hipStream_t rs = ctx->reduce_stream[(&s - ctx->slot) % NREDUCE];
  HIP_TRY(ctx, hipMemsetAsync(s.d_wsums, 0, WSUM_BYTES + 4, rs));
  HIP_TRY(ctx, hipMemsetAsync(s.d_big_queue, 0, 4, rs));
  HIP_TRY(ctx, hipMemsetAsync(s.d_big_queue + BIGQ_COUNTERS, 0, (size_t)STITCH_BLOCKS * 4, rs));
  if (!s.done) HIP_TRY(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  if (!s.smvp_done) HIP_TRY(ctx, hipEventCreateWithFlags(&s.smvp_done, hipEventDisableTiming));
  if (!s.staged) HIP_TRY(ctx, hipEventCreateWithFlags(&s.staged, hipEventDisableTiming));
  if (!s.red0) HIP_TRY(ctx, hipEventCreate(&s.red0));
  if (!s.red1) HIP_TRY(ctx, hipEventCreate(&s.red1));
  for (int i = 0; i < N_MAIN_EVENTS; i++)
    if (!s.ev[i]) HIP_TRY(ctx, hipEventCreate(&s.ev[i]));
  HIP_TRY(ctx, hipEventRecord(s.done, rs));  // the first launch into the slot waits for the memsets through this event
  s.ready = true;
  return MSM_HIP_OK;
}

// make the pools fit a launch of `w_count` local windows (vectors x windows) over n points into slot `s` (not pending)
// (`full_windows`: the windows of one whole MSM in the launch's mode -- the sort arrays are sized for at least that many)
// (`recs_override`: the launch's own count of SMVP lanes, where it is not the one n entries per window give -- wide fixed-base launches)
int ensure_work(msm_hip_ctx* ctx, size_t n, int w_count, int wbits, int full_windows, Slot& s, bool planes, size_t recs_override = 0) {
  int rc;
  if ((rc = setup_slot(ctx, s))) return rc;
  const size_t need_recs = recs_override ? recs_override : (size_t)w_count * chunks_for(n, chunk_len_for(n, w_count));
  const size_t need_entries = stride_for(n) * (size_t)w_count;
  if ((planes || ctx->debug) && need_entries > ctx->cap_planes) {  // digit planes (main stream only, like the sort arrays)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    size_t entries = stride_for(n) * (size_t)full_windows;
    if (entries < need_entries) entries = need_entries;
    if (entries < ctx->cap_entries) entries = ctx->cap_entries;
    ctx->cap_planes = 0;
    if ((rc = dev_alloc(ctx, ctx->d_digits, entries))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_negbits, entries / 64 + 2 * MAXLW))) return rc;
    ctx->cap_planes = entries;
  }
  if (need_entries > ctx->cap_entries || need_recs > ctx->cap_chunk_slot) {
    // growing the context-wide sort arrays (main stream only): nothing may still be running on them
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    size_t entries = stride_for(n) * (size_t)full_windows;  // any single MSM over up to n entries per window
    if (entries < need_entries) entries = need_entries;
    if (entries > ctx->cap_entries) {
      ctx->cap_entries = 0;
      if ((rc = dev_alloc(ctx, ctx->d_tmp_val, entries))) return rc;
      if ((rc = dev_alloc(ctx, ctx->d_tmp_fine, entries))) return rc;
      if ((rc = dev_alloc(ctx, ctx->d_val, entries))) return rc;
      ctx->cap_entries = entries;
    }
    size_t recs = piece_records_for(n);
    if (recs < need_recs) recs = need_recs;
    if (recs > ctx->cap_chunk_slot) {
      ctx->cap_chunk_slot = 0;
      if ((rc = dev_alloc(ctx, ctx->d_chunk_slot, recs))) return rc;
      ctx->cap_chunk_slot = recs;
    }
  }
  if (need_recs > s.cap_recs) {  // this slot's piece arrays (the slot is idle: its previous occupant was collected)
    size_t recs = piece_records_for(n);
    if (recs < need_recs) recs = need_recs;
    s.cap_recs = 0;
    if ((rc = dev_alloc(ctx, s.d_heads, recs * ctx->ops->rec_words))) return rc;
    if ((rc = dev_alloc(ctx, s.d_tails, recs * ctx->ops->rec_words))) return rc;
    s.cap_recs = recs;
  }
  const size_t need_buckets = (size_t)w_count << (wbits - 1);
  if (need_buckets > s.cap_buckets) {  // one MSM's worth (16 x 2^15 at 16 bits) at least; larger launches grow it
    size_t recs = (size_t)NWIN * HALF;
    if (recs < need_buckets) recs = need_buckets;
    s.cap_buckets = 0;
    if ((rc = dev_alloc(ctx, s.d_buckets, recs * ctx->ops->rec_words))) return rc;
    s.cap_buckets = recs;
  }
  if (n >= ctx->fine_hist_min_n && !ctx->d_part_hist) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = dev_alloc(ctx, ctx->d_part_hist, (size_t)MAXLW * NCOARSE * FINE_SPLIT * FINE))) return rc;
  }
  return MSM_HIP_OK;
}

// Window size of a WHOLE-MSM launch (the reference hard-codes c = 16 for n >= 2^16, src/cuzk/msm.rs:79).
// Measured on MI355X (profiles/r02_window_bits_latency.txt): what a small MSM costs is the DEPTH of the bucket reduce (~50
// dependent group additions at ~7 us each), which does not shrink with the bucket count -- so smaller windows, which need more
// windows (19 at 14 bits, 22 at 12), only win while the accumulation itself is negligible: 12 bits up to 2^12 points
// (0.70 vs 0.80 ms latency, 0.34 vs 0.40 ms pipelined), 16 bits from 2^13 up; 14 bits never wins and is kept as an explicit
// choice for ONE MSM per launch (msm_hip_set_window_bits).
// Several whole MSMs per launch (the batch entry points; nvec > 1) are a different regime: 4 - 8 MSMs' bucket sets are stitched and
// reduced side by side, so the work per bucket counts, not the depth of one reduce -- 14 bits (a quarter of the buckets for a
// quarter more additions) wins up to 2^16 points, in both base modes (profiles/r02_window_bits_grouped.txt: 2^12 +41 %, 2^14 +29 %,
// 2^16 +6 % with the endomorphism; +45 % / +50 % / +27 % plain; 16 bits from 2^17 up).
// `nvec` whole MSMs must fit MAXLW local windows.  The window-sharding entry points always use 16-bit windows: their w_begin / w_end
// index the reference's 16 windows.
inline int pick_window_bits(const msm_hip_ctx* ctx, size_t n, int nvec, bool halves = false) {
  static const int forced = [] { const char* e = getenv("MSM_HIP_WINDOW_BITS"); return e ? atoi(e) : 0; }();  // tuning aid
  int bits = ctx->window_bits ? ctx->window_bits : (forced == 12 || forced == 14 || forced == 16 ? forced : 0);
  if (!bits) bits = nvec > 1 ? (n <= ((size_t)1 << 16) ? 14 : 16) : (n <= ((size_t)1 << 12) ? 12 : 16);
  while (bits < 16 && nvec * nwin_of(bits, halves) > MAXLW) bits += 2;
  return bits;
}

// MSM_HIP_DEBUG_SYNC=1 (diagnostic): wait after every kernel of a launch and name it on stderr, so that a device fault is
// pinned to a kernel.  Destroys all overlap; never set for measurements.
inline bool debug_sync() {
  static const bool v = [] { const char* e = getenv("MSM_HIP_DEBUG_SYNC"); return e && e[0] == '1'; }();
  return v;
}
#define AFTER_KERNEL(ctx, name, stream)                                   \
  do {                                                                    \
    if (debug_sync()) {                                                   \
      fprintf(stderr, "[msm_hip] %s ...", name);                          \
      fflush(stderr);                                                     \
      hipError_t e_ = hipStreamSynchronize(stream);                       \
      fprintf(stderr, " %s\n", e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); \
      fflush(stderr);                                                     \
      if (e_ != hipSuccess) {                                             \
        (ctx)->last_hip_error = (int)e_;                                  \
        return MSM_HIP_ERR_HIP;                                           \
      }                                                                   \
    }                                                                     \
  } while (0)

inline unsigned blocks_for(size_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

int err_from_bits(uint32_t bits) {
  if (bits & ERRBIT_NOT_ON_CURVE) return MSM_HIP_ERR_NOT_ON_CURVE;
  if (bits & (ERRBIT_NONCANONICAL | ERRBIT_SCALAR_CARRY)) return MSM_HIP_ERR_NONCANONICAL;
  return MSM_HIP_OK;
}

// Does a launch's second sort pass read digit planes left by the first (k_scatter_planes) instead of the scalars again?  Yes for window
// shares -- at most PLANES_MAX_W of a vector's windows (8 GPUs: 2 of 16, or 1 of 8 half-length windows): 2 B per (input, window) instead
// of 32 B per scalar, and no scalars held in registers.  Not for fixed-base tables (one bucket set per vector) and not while the debug
// read-back wants the planes in its own format.  MSM_HIP_PLANES_MAX_W overrides the limit (0: never; tuning aid).
inline bool use_planes(const msm_hip_ctx* ctx, LaunchMode mode, int w_count_vec, int wbits) {
  static const int max_w = [] { const char* e = getenv("MSM_HIP_PLANES_MAX_W"); return e ? atoi(e) : 8; }();
  static const bool whole = [] { const char* e = getenv("MSM_HIP_PLANES_WHOLE"); return e && e[0] == '1'; }();  // A/B aid: whole MSMs too
  if (mode == MODE_TABLES || mode == MODE_WIDE || ctx->debug) return false;
  if (whole) return true;
  return w_count_vec <= max_w && w_count_vec < nwin_of(wbits, mode == MODE_HALVES);
}

// Enqueue one MSM (windows [w_begin, w_begin + w_count)) into slot `s`.  Window sums (canonical Jacobian bytes) go to
// `wsums_out` (device memory; the slot's own buffer when null); the error word and, if `to_host`, the window sums are
// copied to the slot's pinned buffer.  Returns without waiting.
// (MODE_WIDE: `v_count` != 0 -- a SHARE of the virtual windows, [v_begin, v_begin + v_count) of every vector; its sums are (window sum, plain
//  total) pairs, 2 records per local window.  0: whole MSMs, all 2^(C-16) virtual windows)
// `phase`: 0 the whole launch; 1 only its recode + sort (everything that needs the scalars alone); 2 the rest, from the SMVP on (everything that
// needs the bases) -- the one-shot entry point sorts while the bases are still being uploaded (msm_hip_msm_bn254_g1).  Phase 2 must follow
// phase 1 of the same launch with the same arguments.
int enqueue(msm_hip_ctx* ctx, const uint32_t* d_scalars, size_t n, int w_begin, int w_count_vec, int nvec, int wbits, LaunchMode mode, Slot& s,
            uint32_t* wsums_out, bool to_host, int v_begin = 0, int v_count = 0, int phase = 0) {
  const bool merge = mode == MODE_TABLES, halves = mode == MODE_HALVES, wide = mode == MODE_WIDE;
  const bool pairs = wide && v_count != 0;
  if (wide && !pairs) v_count = wide_vwin_of(ctx->wide_bits);
  const uint32_t half = 1u << (wbits - 1);   // bucket slots per window
  const unsigned ncoarse = half / FINE;      // coarse bins that can hold entries
  // fixed-base tables (`merge`): the w_count_vec windows of a vector feed one bucket set -- one local window of up to
  // n * w_count_vec entries per vector -- whose entries index the tables (window w of point i = record w * n_bases + i)
  // wide tables (`wide`; w_count_vec = the T digits of C bits): the same indexing; each vector's bucket set of 2^(C-1) slots is run as 2^(C-16) local
  // ("virtual") windows of 2^15, into which the entries fall by the top bits of their digit's magnitude (msm_kernels.h: k_count_wide)
  const size_t merge_nb = merge || wide ? ctx->n_bases : 0;
  // endomorphism (`halves`): the recode runs over 2n halves of 16 B (the first pass splits the scalars) against 2n points -- P_i and, n_bases records
  // further on, phi(P_i) -- in half as many windows
  const size_t n_sc = halves ? 2 * n : n;  // inputs of the recode
  const size_t n_entries = merge || wide ? n * (size_t)w_count_vec : n_sc;  // (wide: what ONE virtual window may receive)
  // `nvec` scalar vectors (contiguous, n x 32 B each) share this launch: local window lw = v * w_count_vec + (w - w_begin);
  // everything after the two scalar-reading kernels only sees w_count = nvec * w_count_vec local windows
  const int w_count = merge ? nvec : wide ? nvec * v_count : nvec * w_count_vec;
  hipStream_t st = ctx->stream, rs = ctx->reduce_stream[(&s - ctx->slot) % NREDUCE];
  // a synchronous call with nothing else in flight (msm_hip_run_*: the caller waits for this launch before it issues another): the stitch and
  // the bucket reduce follow the SMVP on the MAIN stream -- no cross-stream hand-off (an event wait costs ~10 us more than an in-stream kernel
  // boundary), and no next launch exists whose sort the separate stream would let overlap.  MSM_HIP_INLINE_REDUCE=0: always the reduce stream.
  static const bool inline_reduce = [] { const char* e = getenv("MSM_HIP_INLINE_REDUCE"); return !e || atoi(e) != 0; }();
  if (inline_reduce && ctx->sync_call) {
    bool others = false;
    for (const Slot& o : ctx->slot) others = others || (&o != &s && o.pending);
    if (!others) rs = st;
  }
  // tiles of scalars for the two global sort passes: >= 2048 scalars each, at most MAX_TILES of them
  // window shares (a few of a scalar's windows per vector): the first pass leaves digit planes, the second reads them (k_scatter_planes)
  const bool planes = use_planes(ctx, mode, w_count_vec, wbits);
  // shares of at most WIDE_SHARE_VWIN_MAX virtual windows of wide tables: the first pass leaves compact lists of the share's entries per sub-tile of
  // LIST_SUB scalars (k_count_wide_list / k_scatter_list); tiles are then whole sub-tiles.  MSM_HIP_WIDE_SHARE_LISTS=0: the two-pass shape (A/B aid)
  static const bool share_lists = [] { const char* e = getenv("MSM_HIP_WIDE_SHARE_LISTS"); return !e || atoi(e) != 0; }();
  const bool share_shape = pairs && v_count <= WIDE_SHARE_VWIN_MAX;
  const bool list_path = share_shape && share_lists;
  const uint32_t tile_unit = list_path ? (uint32_t)LIST_SUB : 256u;
  uint32_t tile_len = 2048;
  if ((n_sc + tile_len - 1) / tile_len > MAX_TILES) tile_len = (uint32_t)((((n_sc + MAX_TILES - 1) / MAX_TILES) + tile_unit - 1) / tile_unit * tile_unit);
  const uint32_t tiles = (uint32_t)((n_sc + tile_len - 1) / tile_len);
  const uint32_t subtiles = (uint32_t)((n_sc + LIST_SUB - 1) / LIST_SUB);
  const WideShape ws = wide ? wide_shape(n, ctx->curve, ctx->wide_bits, w_count) : WideShape{};
  const uint32_t chunk_len = wide ? ws.host_len : chunk_len_for(n_entries, w_count);  // (the longest the device may pick: smvp_chunk_len)
  const uint32_t chunks = wide ? ws.chunks : chunks_for(n_entries, chunk_len);
  const size_t stride = stride_for(n_entries);
  ctx->last_stride = stride;
  uint16_t* digits = ctx->debug && !merge && !wide ? ctx->d_digits : nullptr;
  uint32_t* d_err = reinterpret_cast<uint32_t*>(s.d_wsums + WSUM_BYTES);
  if (!wsums_out) wsums_out = reinterpret_cast<uint32_t*>(s.d_wsums);
  // the SMVP chunk length the device settles on for this launch (k_scatter_coarse -> fine sort, SMVP, stitch): a word of the slot
  uint32_t* d_chunk_len = s.d_big_queue + BIGQ_CHUNK_LEN;

  // the slot's previous occupant (bucket reduce + copies on the reduce stream) must have drained; its error word was
  // re-zeroed at the end of that chain.  Stage events cost a few microseconds of queue time each, so only the ones the
  // current timing level asks for are recorded.
  const int tl = ctx->timing_level;
  auto mark = [&](int i, bool smvp_edge) -> hipError_t {
    if (tl >= 2 || (tl == 1 && smvp_edge)) return hipEventRecord(s.ev[i], st);
    return hipSuccess;
  };
  if (phase != 2) {  // ---- recode + sort
  HIP_TRY(ctx, hipStreamWaitEvent(st, s.done, 0));
  HIP_TRY(ctx, mark(0, false));
  if (ctx->scalar_format == MSM_HIP_SCALARS_MONT256) {  // Montgomery-form scalars: canonical copies first (part of stage 0)
    const size_t count = (size_t)nvec * n;
    hipLaunchKernelGGL(ctx->ops->scalars_from_mont256, dim3(blocks_for(count, 256)), dim3(256), 0, st, d_scalars, ctx->d_scalar_conv, count, d_err);
    AFTER_KERNEL(ctx, "k_scalars_from_mont256", st);
    d_scalars = ctx->d_scalar_conv;
  }
  const unsigned gpos_bytes = (unsigned)(merge ? nvec : nvec * w_count_vec) * NCOARSE * 4;  // k_scatter_coarse's run cursors (dynamic LDS)
#define LAUNCH_BY_WBITS_SW(KERNEL, LDS, SW, ...)                                         \
  do {                                                                                   \
    if (wbits == 16) hipLaunchKernelGGL((KERNEL<16, SW>), dim3(tiles, nvec), dim3(256), LDS, st, __VA_ARGS__); \
    else if (wbits == 14) hipLaunchKernelGGL((KERNEL<14, SW>), dim3(tiles, nvec), dim3(256), LDS, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<12, SW>), dim3(tiles, nvec), dim3(256), LDS, st, __VA_ARGS__); \
  } while (0)
  // first pass: recode + coarse histogram (+ digit planes: 1 = debug read-back, 2 = the second pass reads them).  Endomorphism launches:
  // the same kernel splits every scalar k = k1 + k2 lambda itself and leaves the halves (interleaved: input 2 j = k1 of scalar j, 2 j + 1 =
  // k2; a vector's 2n halves take the room of its n scalars, vector stride n * 8 words either way) for a scalar-reading second pass.
  uint16_t* plane_out = planes ? ctx->d_digits : digits;
  const int plane_mode = planes ? 2 : (digits ? 1 : 0);
  if (wide) {
    const int top_shift = wide_top_shift(ctx->curve, ctx->wide_bits);
#define LAUNCH_COUNT_WIDE(C) hipLaunchKernelGGL(k_count_wide<C>, dim3(tiles, nvec), dim3(256), 0, st, d_scalars, n_sc, tile_len, tiles, nvec, n * 8, ctx->d_counts, d_err, top_shift, v_begin, v_count)
#define LAUNCH_COUNT_WIDE_LIST(C) hipLaunchKernelGGL(k_count_wide_list<C>, dim3(tiles, nvec), dim3(256), 0, st, d_scalars, n_sc, tile_len, tiles, nvec, n * 8, ctx->d_counts, d_err, top_shift, v_begin, v_count, ctx->d_val, ctx->d_list_len, stride, subtiles)
    switch (ctx->wide_bits * 2 + (list_path ? 1 : 0)) {
      case 32: LAUNCH_COUNT_WIDE(16); break;
      case 33: LAUNCH_COUNT_WIDE_LIST(16); break;
      case 34: LAUNCH_COUNT_WIDE(17); break;
      case 35: LAUNCH_COUNT_WIDE_LIST(17); break;
      case 36: LAUNCH_COUNT_WIDE(18); break;
      case 37: LAUNCH_COUNT_WIDE_LIST(18); break;
      case 38: LAUNCH_COUNT_WIDE(19); break;
      case 39: LAUNCH_COUNT_WIDE_LIST(19); break;
      case 40: LAUNCH_COUNT_WIDE(20); break;
      default: LAUNCH_COUNT_WIDE_LIST(20); break;
    }
#undef LAUNCH_COUNT_WIDE
#undef LAUNCH_COUNT_WIDE_LIST
  } else if (halves) {
    hipLaunchKernelGGL(ctx->ops->count_split[wbits == 16 ? 2 : wbits == 14 ? 1 : 0], dim3(tiles, nvec), dim3(256), 0, st, d_scalars, n_sc, tile_len, tiles, w_begin,
                       w_count_vec, nvec, n * 8, ctx->d_counts, plane_out, plane_mode, planes ? ctx->d_negbits : nullptr,
                       planes ? nullptr : ctx->d_halves, d_err, merge_nb);
    d_scalars = ctx->d_halves;
  } else {
    LAUNCH_BY_WBITS_SW(k_count, 0, 8, d_scalars, n_sc, tile_len, tiles, w_begin, w_count_vec, nvec, n * 8, ctx->d_counts, plane_out, plane_mode,
                       (uint64_t*)nullptr, (uint32_t*)nullptr, d_err, merge_nb);
  }
  AFTER_KERNEL(ctx, "k_count", st);
  HIP_TRY(ctx, mark(1, false));
  hipLaunchKernelGGL(k_scan_tiles, dim3(NCOARSE / 4, w_count), dim3(256), 0, st, ctx->d_counts, tiles, ctx->d_bin_total);  // all 128 bins: the scatter scans them
  AFTER_KERNEL(ctx, "k_scan_tiles", st);
  HIP_TRY(ctx, mark(2, false));
  if (wide) {
    const int top_shift = wide_top_shift(ctx->curve, ctx->wide_bits);
    // shares of at most WIDE_SHARE_VWIN_MAX virtual windows: from the first pass's lists, or (MSM_HIP_WIDE_SHARE_LISTS=0) the small shape of the
    // two-pass kernel (four workgroups per CU instead of one)
    if (list_path) {
      hipLaunchKernelGGL(k_scatter_list, dim3(tiles, w_count), dim3(256), 0, st, (const uint32_t*)ctx->d_val, (const uint32_t*)ctx->d_list_len, stride,
                         (uint32_t)(LIST_SUB * wide_tables_of(ctx->wide_bits)), subtiles, n_sc, tile_len, tiles, w_count, ctx->d_counts, ctx->d_bin_total,
                         ctx->d_coarse_ptr, ctx->d_tmp_val, ctx->d_tmp_fine, merge_nb, chunks, chunk_len, d_chunk_len);
    } else {
#define LAUNCH_SCATTER_WIDE_SHAPE(C, SHARE)                                                                                                                         \
  hipLaunchKernelGGL((k_scatter_wide<C, SHARE>), dim3(tiles, nvec), dim3(WideScatterShape<C, SHARE>::THREADS), 0, st, d_scalars, n_sc, stride, tile_len, tiles, nvec, n * 8, ctx->d_counts, ctx->d_bin_total, \
                     ctx->d_coarse_ptr, ctx->d_tmp_val, ctx->d_tmp_fine, merge_nb, chunks, chunk_len, d_chunk_len, top_shift, v_begin, v_count)
#define LAUNCH_SCATTER_WIDE(C)                       \
  do {                                               \
    if (share_shape) LAUNCH_SCATTER_WIDE_SHAPE(C, true); \
    else LAUNCH_SCATTER_WIDE_SHAPE(C, false);        \
  } while (0)
    switch (ctx->wide_bits) {
      case 16: LAUNCH_SCATTER_WIDE(16); break;
      case 17: LAUNCH_SCATTER_WIDE(17); break;
      case 18: LAUNCH_SCATTER_WIDE(18); break;
      case 19: LAUNCH_SCATTER_WIDE(19); break;
      default: LAUNCH_SCATTER_WIDE(20); break;
    }
#undef LAUNCH_SCATTER_WIDE_SHAPE
#undef LAUNCH_SCATTER_WIDE
    }
  } else if (planes) {
    hipLaunchKernelGGL(k_scatter_planes, dim3(tiles), dim3(256), 0, st, ctx->d_digits, halves ? ctx->d_negbits : (const uint64_t*)nullptr, n_sc, stride, tile_len,
                       tiles, w_count, w_count_vec, ctx->d_counts, ctx->d_bin_total, ctx->d_coarse_ptr, ctx->d_tmp_val, ctx->d_tmp_fine,
                       (uint32_t)ctx->n_bases, chunks, chunk_len, d_chunk_len);
  } else if (halves) {
    LAUNCH_BY_WBITS_SW(k_scatter_coarse, gpos_bytes, 4, d_scalars, n_sc, stride, tile_len, tiles, w_begin, w_count_vec, nvec, n * 8, ctx->d_counts, ctx->d_bin_total,
                       ctx->d_coarse_ptr, ctx->d_tmp_val, ctx->d_tmp_fine, merge_nb, (uint32_t)n, (uint32_t)ctx->n_bases, chunks, chunk_len, d_chunk_len);
  } else {
    LAUNCH_BY_WBITS_SW(k_scatter_coarse, gpos_bytes, 8, d_scalars, n_sc, stride, tile_len, tiles, w_begin, w_count_vec, nvec, n * 8, ctx->d_counts, ctx->d_bin_total,
                       ctx->d_coarse_ptr, ctx->d_tmp_val, ctx->d_tmp_fine, merge_nb, (uint32_t)n, 0u, chunks, chunk_len, d_chunk_len);
  }
#undef LAUNCH_BY_WBITS_SW
  AFTER_KERNEL(ctx, "k_scatter_coarse", st);
  HIP_TRY(ctx, mark(3, false));
  const uint32_t* part_hist = nullptr;
  // large n: the sub-range histograms of huge coarse bins are made once, not by every sharer.  (round 5) Not launched at all while uniform scalars
  // cannot fill a coarse bin to three quarters of FINE_BIG -- 2^20 points and below: its 8192 workgroups found nothing to do and took 19 us of every
  // launch's main stream; skewed scalars that make such a bin after all take the sharers' own histograms (k_sort_fine's fallback path)
  // ... ADAPTIVELY (later in round 5): few distinct / small / equal scalars (witness vectors) fill huge bins at any size, and the fallback costs their
  // fine sort 2 - 2.5 x (profiles/r05_skew_hist.txt): k_sort_fine reports a huge bin in the slot's status word, and the 64 launches after such a
  // report run k_fine_hist (a prover's MSMs come in series of like inputs; the first of a series pays the fallback once)
  const bool hist_useful = ctx->fine_hist_min_n != FINE_BIG + 1 || n_entries / ncoarse * 4 > (size_t)FINE_BIG * 3 || ctx->skew_credit > 0;
  if (ctx->skew_credit > 0) ctx->skew_credit--;
  if (n_entries >= ctx->fine_hist_min_n && hist_useful) {
    hipLaunchKernelGGL(k_fine_hist, dim3(ncoarse, w_count, FINE_SPLIT), dim3(256), 0, st, ctx->d_tmp_fine, stride, ctx->d_coarse_ptr,
                       ctx->d_part_hist);
    AFTER_KERNEL(ctx, "k_fine_hist", st);
    part_hist = ctx->d_part_hist;
  }
  hipLaunchKernelGGL(k_sort_fine, dim3(ncoarse, w_count, FINE_SPLIT), dim3(256), 0, st, ctx->d_tmp_val, ctx->d_tmp_fine, stride, ctx->d_coarse_ptr,
                     s.d_col_ptr, ctx->d_val, chunks, d_chunk_len, ctx->d_chunk_slot, part_hist, d_err);
  AFTER_KERNEL(ctx, "k_sort_fine", st);
  if (ctx->debug) {  // deterministic transpose for the stage read-back: every slot's run in ascending order
    hipLaunchKernelGGL(k_order_runs, dim3(blocks_for(n_entries, 256), w_count), dim3(256), 0, st, s.d_col_ptr, ctx->d_val, ctx->d_tmp_val, stride, half);
    hipLaunchKernelGGL(k_copy_runs, dim3(blocks_for(n_entries, 256), w_count), dim3(256), 0, st, s.d_col_ptr, ctx->d_tmp_val, ctx->d_val, stride, half);
    AFTER_KERNEL(ctx, "k_order_runs", st);
  }
  }  // ---- (recode + sort)
  if (phase == 1) {
    HIP_TRY(ctx, hipGetLastError());
    return MSM_HIP_OK;
  }
  // the SMVP's own begin / end timestamps: attached to its dispatch (hipExtLaunchKernelGGL) instead of two event packets around it -- a
  // packet between two kernels costs a few microseconds of queue time, and these two sat between every launch's sort and its SMVP and
  // between the SMVP and the next launch (bench.py times every launch's SMVP for the roofline figure).  Level 2 keeps the packets: its
  // stage boundaries are read as differences of consecutive events.
  static const unsigned smvp_lds_pad = [] { const char* e = getenv("MSM_HIP_SMVP_LDS_PAD"); const long v = e ? atol(e) : 0; return v > 0 && v <= 65536 ? (unsigned)v : 0u; }();
  HIP_TRY(ctx, tl >= 2 ? hipEventRecord(s.ev[4], st) : hipSuccess);
  hipExtLaunchKernelGGL(ctx->ops->smvp_chunks, dim3((chunks + 255) / 256, w_count), dim3(256), smvp_lds_pad, st, tl == 1 ? s.ev[4] : nullptr,
                        tl == 1 ? s.ev[5] : nullptr, 0, (const uint32_t*)(ctx->d_bases + s.base_off * 2 * (size_t)ctx->ops->coord_words), (const uint32_t*)s.d_col_ptr, (const uint32_t*)ctx->d_val, stride, chunks,
                        (const uint32_t*)d_chunk_len, (const uint32_t*)ctx->d_chunk_slot, s.d_buckets, s.d_heads, s.d_tails, half);
  AFTER_KERNEL(ctx, "k_smvp_chunks", st);
  HIP_TRY(ctx, tl >= 2 ? hipEventRecord(s.ev[5], st) : hipSuccess);
  if (rs != st) HIP_TRY(ctx, hipEventRecord(s.smvp_done, st));

  // stitch + bucket reduce on the slot's reduce stream: few waves, long dependent chains
  if (rs != st) HIP_TRY(ctx, hipStreamWaitEvent(rs, s.smvp_done, 0));
  hipLaunchKernelGGL(ctx->ops->smvp_stitch, dim3(half / 256, w_count), dim3(256), 0, rs, s.d_col_ptr, chunks, d_chunk_len, s.d_heads, s.d_tails,
                     s.d_buckets, s.d_big_queue);
  AFTER_KERNEL(ctx, "k_smvp_stitch", rs);
  hipLaunchKernelGGL(ctx->ops->smvp_stitch_big, dim3(STITCH_BLOCKS), dim3(256), 0, rs, s.d_col_ptr, chunks, s.d_heads, s.d_tails, s.d_buckets,
                     s.d_big_queue, half);
  AFTER_KERNEL(ctx, "k_smvp_stitch_big", rs);
  if (tl >= 2) {
    HIP_TRY(ctx, hipEventRecord(s.ev[6], rs));
    HIP_TRY(ctx, hipEventRecord(s.red0, rs));
  }
  uint32_t* d_rows = s.d_partials;
  uint32_t* d_cols = d_rows + (size_t)MAXLW * 256 * ctx->ops->xyzz_words;
  uint32_t* d_parts = d_cols + (size_t)MAXLW * 256 * ctx->ops->xyzz_words;
  static const int force_logr = [] { const char* e = getenv("MSM_HIP_BPR_LOGR"); return e ? atoi(e) : 0; }();  // tuning aid
  // serial run per thread before the LDS tree: 16 buckets when many windows are reduced at once (fewest wave-additions),
  // 4 for a few windows (shallowest); measured optimum for 16 and for 2 windows respectively
#define ROWCOL(LOG_R, LOG_ROWS) \
  hipLaunchKernelGGL(ctx->ops->rowcol_##LOG_R##_##LOG_ROWS, dim3(bpr_rowcol_blocks<LOG_R, LOG_ROWS>(), w_count), dim3(256), 0, rs, s.d_buckets, d_rows, d_cols)
  if (wbits == 16) {
    // one small MSM alone in its launch (8 half-length windows, up to 2^18 points): 8 buckets per thread -- its latency is what counts
    // (-4 % at 2^16, -2.6 % at 2^18; the same setting costs grouped launches 3 - 16 % and a pipelined 2^20 MSM 0.5 %)
    // ... and so is a larger one that has the GPU to itself: no other result slot of the context is in flight when it is launched
    // (latency 1.81 -> 1.77 ms at 2^20; in a pipeline only the very first launch is alone: throughput unchanged)
    bool alone = nvec == 1 && w_count == 8;
    for (const Slot& o : ctx->slot) alone = alone && (&o == &s || !o.pending);
    const bool small_single = nvec == 1 && w_count == 8 && (n_entries <= ((size_t)1 << 19) || alone);
    if (force_logr == 4 || (force_logr == 0 && w_count >= 8 && !small_single)) ROWCOL(4, 8);
    else if (force_logr == 3 || (force_logr == 0 && small_single)) ROWCOL(3, 8);
    else ROWCOL(2, 8);
  } else if (wbits == 14) {  // 64 rows x 128 columns
    if (force_logr == 4 || (force_logr == 0 && w_count > 2 * nwin_of(14))) ROWCOL(4, 6);
    else ROWCOL(2, 6);
  } else {  // 16 rows x 128 columns
    ROWCOL(2, 4);
  }
#undef ROWCOL
  AFTER_KERNEL(ctx, "k_bpr_rowcol", rs);
  // A launch that carries ONE MSM whose sums the host combines anyway (finish): the narrow end of the reduction -- ~25 dependent group
  // operations at ~7 us each in k_bpr_w256 / k_bpr_final -- is replaced by 16 independent masked tree sums per window (k_bpr_planes, 8 deep)
  // and 29 operations per window on the host (0.3 us each).  Sums that stay on the device (window shards for the gather), grouped launches
  // (their host thread is on the critical path: several MSMs' worth of host work per launch) and debug read-backs get finished sums.
  // (a wide fixed-base launch always leaves the plane sums -- its finish needs every virtual window's plain total, which is one of them --: launch_impl
  //  admits it only as whole MSMs whose sums go to the host, at most 24 virtual windows together)
  const bool parts_mode = to_host && !pairs && (nvec == 1 || wide) && (!ctx->debug || wide) && wsums_out == reinterpret_cast<uint32_t*>(s.d_wsums) && w_count <= 24;
  // the kernel that ends the chain writes the launch's error word into the slot's pinned buffer itself and clears it (no copy, no fill); a
  // single MSM's bit-plane sums go to the pinned buffer directly as well (12 KB of stores over the host link instead of a copy behind the kernel)
  uint32_t* h_err = reinterpret_cast<uint32_t*>(s.h_wsums + WSUM_BYTES);
  if (parts_mode) {
    hipLaunchKernelGGL(ctx->ops->bpr_planes, dim3(PLANES_PER_WINDOW, w_count), dim3(256), 0, rs, d_rows, d_cols, reinterpret_cast<uint32_t*>(s.h_wsums),
                       (int)(half / BPR_COLS), s.d_big_queue, d_err, h_err);
    AFTER_KERNEL(ctx, "k_bpr_planes", rs);
  } else if (ctx->ops->use_w256) {
    hipLaunchKernelGGL(ctx->ops->bpr_w256, dim3(2, w_count), dim3(256), 0, rs, d_rows, d_cols, d_parts, (int)(half / BPR_COLS));
    AFTER_KERNEL(ctx, "k_bpr_w256", rs);
    hipLaunchKernelGGL(ctx->ops->bpr_final, dim3(1), dim3(64), 0, rs, d_parts, w_count, wsums_out, s.d_big_queue, d_err, h_err, pairs ? 1 : 0);
    AFTER_KERNEL(ctx, "k_bpr_final", rs);
  } else {  // a field too wide for k_bpr_w256's LDS footprint (BLS12-381): the same bit-plane sums, finished on the device
    hipLaunchKernelGGL(ctx->ops->bpr_planes_xyzz, dim3(PLANES_PER_WINDOW, w_count), dim3(256), 0, rs, d_rows, d_cols, d_parts, (int)(half / BPR_COLS),
                       s.d_big_queue, d_err, h_err);
    AFTER_KERNEL(ctx, "k_bpr_planes<xyzz>", rs);
    hipLaunchKernelGGL(ctx->ops->bpr_final_planes, dim3(1), dim3(64), 0, rs, d_parts, w_count, wsums_out, s.d_big_queue, d_err, h_err, pairs ? 1 : 0);
    AFTER_KERNEL(ctx, "k_bpr_final_planes", rs);
  }
  if (tl >= 2) HIP_TRY(ctx, hipEventRecord(s.red1, rs));
  if (to_host && !parts_mode) HIP_TRY(ctx, hipMemcpyAsync(s.h_wsums, wsums_out, (size_t)w_count * (pairs ? 2 : 1) * ctx->jb, hipMemcpyDeviceToHost, rs));
  HIP_TRY(ctx, hipEventRecord(s.done, rs));
  HIP_TRY(ctx, hipGetLastError());

  s.w_begin = w_begin;
  s.w_count = w_count_vec;
  s.nvec = nvec;
  s.wbits = wbits;
  s.merged = merge;
  s.halves = halves;
  s.wide_bits = wide ? ctx->wide_bits : 0;
  s.parts = parts_mode;
  s.pairs = pairs;
  if (pairs) {
    s.w_begin = v_begin;
    s.w_count = v_count;
  }
  s.n = n;
  s.timed = tl >= 1;
  s.timing_level = tl;
  s.pending = true;
  s.to_host = to_host;
  ctx->last_n = n_sc;
  ctx->last_w_count = w_count;
  ctx->last_wbits = wbits;
  ctx->last_slot = (int)(&s - ctx->slot);
  ctx->last_has_digits = digits != nullptr;
  return MSM_HIP_OK;
}

// wait for slot `s`, fetch its error word and stage times
int wait_slot(msm_hip_ctx* ctx, Slot& s) {
  HIP_TRY(ctx, hipEventSynchronize(s.done));
  s.pending = false;
  if (s.timed) {
    s.timed = false;
    for (int i = 0; i < 8; i++) ctx->stage_ms[i] = 0.0f;
    if (s.timing_level >= 2) {
      for (int i = 0; i < 6; i++) HIP_TRY(ctx, hipEventElapsedTime(&ctx->stage_ms[i], s.ev[i], s.ev[i + 1]));
      HIP_TRY(ctx, hipEventElapsedTime(&ctx->stage_ms[6], s.red0, s.red1));
      HIP_TRY(ctx, hipEventElapsedTime(&ctx->stage_ms[7], s.ev[0], s.red1));
    } else {
      HIP_TRY(ctx, hipEventElapsedTime(&ctx->stage_ms[4], s.ev[4], s.ev[5]));
    }
  }
  uint32_t bits;
  memcpy(&bits, s.h_wsums + WSUM_BYTES, 4);
  if (bits & INFOBIT_HUGE_BIN) ctx->skew_credit = 64;  // (see the launch of k_fine_hist)
  return err_from_bits(bits);
}

// after a failed batch: collect every slot that is still pending so that the context stays usable
void drain_slots(msm_hip_ctx* ctx) {
  for (Slot& s : ctx->slot)
    if (s.pending) (void)wait_slot(ctx, s);
}

constexpr size_t MAX_POINTS = (size_t)1 << 28;  // point indices carry the digit sign in bit 31; 2^28 keeps every per-window offset in u32

int check_run_args(msm_hip_ctx* ctx, const void* scalars, size_t n) {
  if (!ctx || (!scalars && n) || n > MAX_POINTS) return MSM_HIP_ERR_INVALID_ARG;
  if (ctx->n_bases == 0 && n) return MSM_HIP_ERR_NO_BASES;
  if (n > ctx->n_bases) return MSM_HIP_ERR_INVALID_ARG;
  return MSM_HIP_OK;
}

// room for n bases; no run may still be reading the old ones (the SMVP on the main stream)
constexpr size_t MAX_PRECOMPUTE_POINTS = (size_t)1 << 24;  // 16 tables: 16 GiB, and table indices stay below 2^28

// The mode a base set is held in.  Asked for explicitly (tables, endomorphism images, or MSM_HIP_BASES_PLAIN: the reference's 16 windows
// over n points), or -- none of the three -- the fastest the curve has: the drop-in call shape (flags = 0; msm_hip_msm_bn254_g1 ≙ compute_msm,
// src/cuzk/msm.rs:75-94) runs the mode the headline figure is measured in (656 vs 701 - 714 MSM/s at 2^20 in round 3, when it did not).
constexpr uint32_t BASE_FLAGS_ALL = MSM_HIP_CHECK_ON_CURVE | MSM_HIP_BASES_MONT256 | MSM_HIP_BASES_PRECOMPUTE | MSM_HIP_BASES_ENDOMORPHISM | MSM_HIP_BASES_PLAIN |
                                    MSM_HIP_BASES_PRECOMPUTE_WIDE;
constexpr size_t MAX_WIDE_POINTS = (size_t)1 << 24;  // 13 tables of 20-bit digits: 13 GiB; sort arrays of 16 x 13 n entries: 31 GiB
inline uint32_t resolve_base_flags(const msm_hip_ctx* ctx, size_t n, uint32_t flags) {
  static const bool auto_endo = [] { const char* e = getenv("MSM_HIP_BASES_AUTO"); return !e || atoi(e) != 0; }();  // MSM_HIP_BASES_AUTO=0: flags = 0 means plain (rounds 1 - 3)
  if (flags & (MSM_HIP_BASES_PRECOMPUTE | MSM_HIP_BASES_PRECOMPUTE_WIDE | MSM_HIP_BASES_ENDOMORPHISM | MSM_HIP_BASES_PLAIN)) return flags;
  // ... on the curves of prime order only: phi(P) = lambda P holds on the subgroup of order r, and a base set of a curve with a cofactor
  // (BLS12-381, the G2 twists) may hold points outside it, for which the plain MSM is still defined -- there the mode stays an opt-in
  const bool prime_order = ctx->curve == MSM_HIP_CURVE_BN254_G1 || ctx->curve == MSM_HIP_CURVE_GRUMPKIN || ctx->curve == MSM_HIP_CURVE_PALLAS ||
                           ctx->curve == MSM_HIP_CURVE_VESTA;
  if (auto_endo && prime_order && ctx->ops->glv && n <= MAX_POINTS / 2) return flags | MSM_HIP_BASES_ENDOMORPHISM;
  return flags;
}

int reserve_bases(msm_hip_ctx* ctx, size_t n, uint32_t flags) {
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const bool wide = (flags & MSM_HIP_BASES_PRECOMPUTE_WIDE) != 0;
  const bool tables = (flags & MSM_HIP_BASES_PRECOMPUTE) != 0 || wide, endo = (flags & MSM_HIP_BASES_ENDOMORPHISM) != 0;
  if (n > MAX_POINTS || (tables && n > MAX_PRECOMPUTE_POINTS) || (wide && n > MAX_WIDE_POINTS) || (endo && n > MAX_POINTS / 2) || (tables && endo))
    return MSM_HIP_ERR_INVALID_ARG;
  if ((flags & ~BASE_FLAGS_ALL) || ((flags & MSM_HIP_BASES_PLAIN) && (tables || endo)) || (wide && (flags & MSM_HIP_BASES_PRECOMPUTE)))
    return MSM_HIP_ERR_INVALID_ARG;
  if ((endo && !ctx->ops->glv) || (tables && !ctx->ops->precompute_tables)) return MSM_HIP_ERR_INVALID_ARG;
  if (wide && pick_wide_bits(ctx, n) < 0) return MSM_HIP_ERR_INVALID_ARG;  // (msm_hip_set_wide_bits asked for a width that cannot hold this curve's scalars)
  ctx->n_bases = 0;
  ctx->precomputed = false;
  ctx->wide_bits = 0;
  ctx->endo = false;
  const size_t records = wide ? n * (size_t)wide_tables_of(pick_wide_bits(ctx, n)) : tables ? n * NWIN : endo ? 2 * n : n;
  if (records > ctx->cap_bases) {
    ctx->cap_bases = 0;
    int rc = dev_alloc(ctx, ctx->d_bases, records * 2 * (size_t)ctx->ops->coord_words);
    if (rc) return rc;
    ctx->cap_bases = records;
  }
  return MSM_HIP_OK;
}

// wire bytes at d_xy (may be ctx->d_bases itself: the conversion is element-wise) -> resident Montgomery bases
int set_bases_from_device(msm_hip_ctx* ctx, const uint32_t* d_xy, size_t n, uint32_t flags) {
  if (n == 0) return MSM_HIP_OK;
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_err, 0, 4, ctx->stream));
  hipLaunchKernelGGL(ctx->ops->convert_points, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_xy, ctx->d_bases, n, flags, ctx->d_err);
  HIP_TRY(ctx, hipGetLastError());
  uint32_t bits = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&bits, ctx->d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  int rc = err_from_bits(bits);
  if (rc) return rc;
  if (flags & (MSM_HIP_BASES_PRECOMPUTE | MSM_HIP_BASES_PRECOMPUTE_WIDE)) {  // tables 1 .. 15 behind the plain set: T_w[i] = 2^(16 w) P_i (wide: 1 .. 13, 2^(19 w) P_i, the last one top_shift doublings short)
    const bool wide = (flags & MSM_HIP_BASES_PRECOMPUTE_WIDE) != 0;
    const int wb = wide ? pick_wide_bits(ctx, n) : 0;
    hipLaunchKernelGGL(ctx->ops->precompute_tables, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_bases, n, n, wide ? wide_tables_of(wb) : (int)NWIN,
                       wide ? wb : (int)WBITS, wide ? wb - wide_top_shift(ctx->curve, wb) : (int)WBITS);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->precomputed = !wide;
    ctx->wide_bits = wb;
  }
  if (flags & MSM_HIP_BASES_ENDOMORPHISM) {  // phi(P_i) = (beta x_i, y_i) behind the plain set
    hipLaunchKernelGGL(ctx->ops->endo_points, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_bases, n, (size_t)0, n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->endo = true;
  }
  ctx->n_bases = n;
  return MSM_HIP_OK;
}

}  // namespace

namespace {
// MSMs per launch of the batch runners: small MSMs cannot fill the GPU one at a time (kernel latencies dominate below
// ~2^19 points), so up to MAXLW / NWIN = 4 of them -- at most about 2^20 points together -- share one kernel sequence
size_t batch_group(msm_hip_ctx* ctx, size_t n, size_t batch) {
  // 4 at 16 bits, 3 at 14, 2 at 12 (8 / 6 / 5 with the endomorphism's half-length scalars); with fixed-base tables every MSM is one local window
  // (window size of a grouped launch: pick_window_bits with nvec > 1)
  // (wide tables: an MSM is 2^(C-16) local windows, and its launch leaves bit-plane sums for at most 24 of them: 24 / 12 / 3 / 1 MSMs at 16 / 17 / 19 / 20 bits)
  const size_t fit = ctx->wide_bits ? (size_t)(24 / wide_vwin_of(ctx->wide_bits)) : ctx->precomputed ? (size_t)MAXLW : (size_t)(MAXLW / nwin_of(pick_window_bits(ctx, n, 2, ctx->endo), ctx->endo));
  size_t g = n ? ((size_t)1 << 20) / n : 1;
  if (g > fit) g = fit;
  if (g > batch) g = batch;
  return g ? g : 1;
}

// software pipeline over the result slots: the host combine of group j overlaps the device work of groups j+1 .. j+2.
// `stage` (may be null) copies group j's scalars to the device and returns their device address.
template <typename Stage>
int run_batch_groups(msm_hip_ctx* ctx, size_t n, size_t batch, uint8_t* out_xyz, Stage stage) {
  const size_t g = batch_group(ctx, n, batch);
  const size_t groups = (batch + g - 1) / g;
  constexpr size_t DEPTH = NSLOT - 1;
  int rc = MSM_HIP_OK;
  for (size_t j = 0; j < groups + DEPTH; j++) {
    if (j >= DEPTH) {
      const size_t k = j - DEPTH;
      if ((rc = msm_hip_finish_batch(ctx, (int)(k % NSLOT), out_xyz + ctx->jb * k * g))) break;
    }
    if (j < groups) {
      const size_t first = j * g, count = first + g <= batch ? g : batch - first;
      const void* dev = nullptr;
      if ((rc = stage(j, first, count, &dev))) break;
      if ((rc = msm_hip_launch_windows_batch_device(ctx, dev, n, (int)count, 0, NWIN, (int)(j % NSLOT), nullptr))) break;
    }
  }
  if (rc) drain_slots(ctx);
  return rc;
}
}  // namespace

extern "C" {

int msm_hip_abi_version(void) { return 7; }  // 7 (round 5): curve-neutral names, virtual-window launches + pair combine, msm_hip_msm_curve, msm_hip_mgpu_set_wide_bits

const char* msm_hip_strerror(int code) {
  switch (code) {
    case MSM_HIP_OK: return "ok";
    case MSM_HIP_ERR_NO_DEVICE: return "no usable HIP device";
    case MSM_HIP_ERR_INVALID_ARG: return "invalid argument";
    case MSM_HIP_ERR_OUT_OF_MEMORY: return "out of device memory";
    case MSM_HIP_ERR_NONCANONICAL: return "non-canonical field element in input";
    case MSM_HIP_ERR_NOT_ON_CURVE: return "base point not on the curve";
    case MSM_HIP_ERR_NO_BASES: return "bases not set";
    case MSM_HIP_ERR_HIP: return "HIP runtime error";
    case MSM_HIP_ERR_SLOT_BUSY: return "result slot still holds an unfinished MSM";
    default: return "unknown error";
  }
}

int msm_hip_last_hip_error(msm_hip_ctx* ctx) { return ctx ? ctx->last_hip_error : 0; }
void* msm_hip_stream(msm_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int msm_hip_ctx_create(msm_hip_ctx** out, int device_id) { return msm_hip_ctx_create_curve(out, device_id, MSM_HIP_CURVE_BN254_G1); }

int msm_hip_ctx_curve(const msm_hip_ctx* ctx) { return ctx ? ctx->curve : MSM_HIP_ERR_INVALID_ARG; }

int msm_hip_ctx_create_curve(msm_hip_ctx** out, int device_id, int curve) {
  if (!out) return MSM_HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSM_HIP_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= count) return MSM_HIP_ERR_INVALID_ARG;
  DeviceGuard guard(device_id);
  if (!guard.ok) return MSM_HIP_ERR_NO_DEVICE;
  msm_hip_ctx* ctx = new (std::nothrow) msm_hip_ctx();
  if (!ctx) return MSM_HIP_ERR_OUT_OF_MEMORY;
  ctx->device = device_id;
  ctx->curve = curve;
  ctx->ops = curve_ops(curve);
  ctx->cb = 4 * (size_t)ctx->ops->coord_words;
  ctx->pb = 2 * ctx->cb;
  ctx->jb = 3 * ctx->cb;
  if (const char* e = getenv("MSM_HIP_FINE_HIST_MIN_LOGN")) {  // tuning aid
    const int l = atoi(e);
    if (l >= 0 && l < 40) ctx->fine_hist_min_n = (size_t)1 << l;
  }
  int rc = MSM_HIP_OK;
  auto fail = [&](int code) {
    msm_hip_ctx_destroy(ctx);
    return code;
  };
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(MSM_HIP_ERR_NO_DEVICE);
  // The reduce streams run at the device's highest stream priority: the previous launch's stitch and bucket reduce then finish beside the
  // next launch's sort (latency-bound kernels that leave the multiplier idle) instead of trailing into its SMVP, which they slow down
  // (2^20, one GPU, five same-box pairs: 1.4025 vs 1.4231 ms per MSM, SMVP 0.976 vs 0.989 ms; a rank's window shares and 2^16: within the
  // noise; the opposite assignment -- main stream high, reduce low -- costs 3 %).  MSM_HIP_REDUCE_PRIORITY=0: plain streams.
  static const bool reduce_high = [] { const char* e = getenv("MSM_HIP_REDUCE_PRIORITY"); return !e || atoi(e) != 0; }();
  int prio_least = 0, prio_greatest = 0;
  if (reduce_high && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = prio_least = 0;
  for (int k = 0; k < NREDUCE; k++) {
    if (reduce_high && prio_greatest != prio_least &&
        hipStreamCreateWithPriority(&ctx->reduce_stream[k], hipStreamNonBlocking, prio_greatest) == hipSuccess)
      continue;
    (void)hipGetLastError();
    ctx->reduce_stream[k] = nullptr;
    if (hipStreamCreateWithFlags(&ctx->reduce_stream[k], hipStreamNonBlocking) != hipSuccess) return fail(MSM_HIP_ERR_NO_DEVICE);
  }
  if (hipEventCreateWithFlags(&ctx->input_ready, hipEventDisableTiming) != hipSuccess) return fail(MSM_HIP_ERR_HIP);
  if ((rc = dev_alloc(ctx, ctx->d_counts, (size_t)MAXLW * MAX_TILES * NCOARSE))) return fail(rc);
  if ((rc = dev_alloc(ctx, ctx->d_bin_total, (size_t)MAXLW * NCOARSE))) return fail(rc);
  if ((rc = dev_alloc(ctx, ctx->d_coarse_ptr, (size_t)MAXLW * (NCOARSE + 1)))) return fail(rc);
  if ((rc = dev_alloc(ctx, ctx->d_err, 1))) return fail(rc);
  // result slots (buckets, piece arrays, events) are set up by the first launch that uses them: setup_slot / ensure_work
  *out = ctx;
  return MSM_HIP_OK;
}

void msm_hip_ctx_destroy(msm_hip_ctx* ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  for (hipStream_t r : ctx->reduce_stream)
    if (r) (void)hipStreamSynchronize(r);
  void* bufs[] = {ctx->d_list_len, ctx->d_bases,   ctx->d_halves, ctx->d_batch_stage, ctx->d_scalar_conv, ctx->d_part_hist, ctx->d_digits, ctx->d_negbits, ctx->d_counts,     ctx->d_bin_total, ctx->d_coarse_ptr,
                  ctx->d_tmp_val, ctx->d_tmp_fine, ctx->d_val,    ctx->d_chunk_slot, ctx->d_err,       ctx->d_stage};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (int k = 0; k < NSLOT; k++) {
    Slot& s = ctx->slot[k];
    if (s.h_wsums) (void)hipHostFree(s.h_wsums);
    void* sbufs[] = {s.d_wsums, s.d_buckets, s.d_partials, s.d_col_ptr, s.d_heads, s.d_tails, s.d_big_queue, s.d_host_scalars};
    for (void* b : sbufs)
      if (b) (void)hipFree(b);
    hipEvent_t evs[] = {s.done, s.smvp_done, s.staged, s.red0, s.red1};
    for (hipEvent_t e : evs)
      if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < N_MAIN_EVENTS; i++)
      if (s.ev[i]) (void)hipEventDestroy(s.ev[i]);
  }
  if (ctx->input_ready) (void)hipEventDestroy(ctx->input_ready);
  if (ctx->bases_ready) (void)hipEventDestroy(ctx->bases_ready);
  for (hipEvent_t e : ctx->chunk_landed)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  for (hipStream_t r : ctx->reduce_stream)
    if (r) (void)hipStreamDestroy(r);
  delete ctx;
}

int msm_hip_wait_stream(msm_hip_ctx* ctx, void* producer_stream) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipEventRecord(ctx->input_ready, static_cast<hipStream_t>(producer_stream)));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->input_ready, 0));
  return MSM_HIP_OK;
}

int msm_hip_set_bases_device(msm_hip_ctx* ctx, const void* xy_dev, size_t n, uint32_t flags) {
  if (!ctx || (!xy_dev && n)) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  flags = resolve_base_flags(ctx, n, flags);
  int rc = reserve_bases(ctx, n, flags);
  if (rc) return rc;
  return set_bases_from_device(ctx, static_cast<const uint32_t*>(xy_dev), n, flags);
}

int msm_hip_set_bases(msm_hip_ctx* ctx, const uint8_t* xy_host, size_t n, uint32_t flags) {
  if (!ctx || (!xy_host && n)) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  flags = resolve_base_flags(ctx, n, flags);
  int rc = reserve_bases(ctx, n, flags);
  if (rc) return rc;
  // the wire bytes land in the bases array itself and are converted in place (same 64 B per point): no staging buffer
  if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bases, xy_host, n * ctx->pb, hipMemcpyHostToDevice, ctx->stream));
  return set_bases_from_device(ctx, ctx->d_bases, n, flags);
}

}  // extern "C"

namespace {
// windows [w_begin, w_end) -- in units of `wbits`-bit windows -- of `nvec` scalar vectors into `slot`
// (MODE_WIDE, v_count != 0: a SHARE of the wide tables' virtual windows -- [v_begin, v_begin + v_count) of every vector; its sums are (window sum,
//  plain total) pairs: 2 records per local window, to `window_sums_dev` or, when null, to the slot's pinned buffer)
int launch_impl(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int w_begin, int w_end, int wbits, int slot,
                void* window_sums_dev, LaunchMode mode = MODE_PLAIN, int v_begin = 0, int v_count = 0, int phase = 0) {
  // (MODE_WIDE: called with wbits = 19 and all 14 windows; everything behind the recode sees 8 local windows of 16 bits)
  const bool merge = mode == MODE_TABLES, halves = mode == MODE_HALVES, wide = mode == MODE_WIDE;
  const bool pairs = wide && v_count != 0;
  int rc = check_run_args(ctx, scalars_dev, n);
  const size_t base_off = ctx ? ctx->launch_base_off : 0;
  if (ctx && phase != 1) ctx->launch_base_off = 0;  // (a two-phase launch passes here twice)
  if (rc) return rc;
  if (base_off + n > ctx->n_bases) return MSM_HIP_ERR_INVALID_ARG;
  if (slot < 0 || slot >= NSLOT || w_begin < 0 || w_end > nwin_of(wbits, halves) || w_begin >= w_end) return MSM_HIP_ERR_INVALID_ARG;
  if (wide && (nvec < 1 || w_begin != 0 || wbits != ctx->wide_bits || w_end != wide_tables_of(wbits))) return MSM_HIP_ERR_INVALID_ARG;
  if (wide && !pairs && (nvec * wide_vwin_of(ctx->wide_bits) > 24 || window_sums_dev)) return MSM_HIP_ERR_INVALID_ARG;
  if (pairs && (v_begin < 0 || v_count < 0 || v_begin + v_count > wide_vwin_of(ctx->wide_bits))) return MSM_HIP_ERR_INVALID_ARG;
  if (!wide && v_count) return MSM_HIP_ERR_INVALID_ARG;
  const int WIDE_VWIN = wide ? (pairs ? v_count : wide_vwin_of(ctx->wide_bits)) : 0;
  if (wide) wbits = WBITS;
  const int w_count = w_end - w_begin;
  const int w_local = merge ? nvec : wide ? nvec * WIDE_VWIN : nvec * w_count;  // bucket sets of the launch
  if (nvec < 1 || w_local > MAXLW) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  Slot& s = ctx->slot[slot];
  if (s.pending) return MSM_HIP_ERR_SLOT_BUSY;  // its result was never collected (msm_hip_finish / msm_hip_slot_sync)
  if ((rc = setup_slot(ctx, s))) return rc;
  s.n = n;
  s.base_off = base_off;
  s.w_begin = w_begin;
  s.w_count = w_count;
  s.nvec = nvec;
  s.wbits = wbits;
  s.merged = merge;
  s.halves = halves;
  s.wide_bits = wide ? ctx->wide_bits : 0;
  s.parts = false;
  s.pairs = pairs;
  if (pairs) {
    s.w_begin = v_begin;
    s.w_count = v_count;
  }
  s.to_host = window_sums_dev == nullptr;
  if (n == 0) {  // identity window sums, nothing to compute
    s.pending = true;
    s.timed = false;
    memset(s.h_wsums, 0, WSUM_BYTES + 4);
    if (window_sums_dev) {
      HIP_TRY(ctx, hipMemsetAsync(window_sums_dev, 0, (size_t)w_local * (pairs ? 2 : 1) * ctx->jb, ctx->reduce_stream[slot % NREDUCE]));
      HIP_TRY(ctx, hipEventRecord(s.done, ctx->reduce_stream[slot % NREDUCE]));
    }
    return MSM_HIP_OK;
  }
  if (wide) {
    const WideShape ws = wide_shape(n, ctx->curve, ctx->wide_bits, w_local);
    if ((rc = ensure_work(ctx, ws.worst, w_local, wbits, wide_vwin_of(ctx->wide_bits), s, false, (size_t)w_local * ws.chunks))) return rc;
    const size_t need_len = pairs ? (size_t)w_local * ((n + LIST_SUB - 1) / LIST_SUB) : 0;  // list lengths of a share's first pass
    if (need_len > ctx->cap_list_len) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      ctx->cap_list_len = 0;
      if ((rc = dev_alloc(ctx, ctx->d_list_len, need_len))) return rc;
      ctx->cap_list_len = need_len;
    }
  } else if ((rc = ensure_work(ctx, merge ? n * (size_t)w_count : halves ? 2 * n : n, w_local, wbits,
                               merge ? 1 : halves ? nwin_of(wbits, true) : NWIN, s, use_planes(ctx, mode, w_count, wbits)))) return rc;
  if (halves && !use_planes(ctx, mode, w_count, wbits) && (size_t)nvec * n > ctx->cap_halves) {  // (the halves as an array: only when the second pass reads them)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cap_halves = 0;
    if ((rc = dev_alloc(ctx, ctx->d_halves, (size_t)nvec * n * 8))) return rc;
    ctx->cap_halves = (size_t)nvec * n;
  }
  if (ctx->scalar_format == MSM_HIP_SCALARS_MONT256 && (size_t)nvec * n > ctx->cap_scalar_conv) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cap_scalar_conv = 0;
    if ((rc = dev_alloc(ctx, ctx->d_scalar_conv, (size_t)nvec * n * 8))) return rc;
    ctx->cap_scalar_conv = (size_t)nvec * n;
  }
  return enqueue(ctx, static_cast<const uint32_t*>(scalars_dev), n, w_begin, w_count, nvec, wbits, mode, s,
                 static_cast<uint32_t*>(window_sums_dev), window_sums_dev == nullptr, v_begin, v_count, phase);
}
}  // namespace

extern "C" {

int msm_hip_launch_windows_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int w_begin, int w_end,
                                              int slot, void* window_sums_dev) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  // whole MSMs whose sums stay in the slot (finish / finish_batch combines them): the window size follows n
  if (w_begin == 0 && w_end == NWIN && window_sums_dev == nullptr && nvec >= 1 && ctx->wide_bits && n > 0)
    return launch_impl(ctx, scalars_dev, n, nvec, 0, wide_tables_of(ctx->wide_bits), ctx->wide_bits, slot, nullptr, MODE_WIDE);  // wide fixed-base tables (one MSM per launch)
  if (w_begin == 0 && w_end == NWIN && window_sums_dev == nullptr && nvec >= 1 && ctx->precomputed && n > 0)
    return launch_impl(ctx, scalars_dev, n, nvec, 0, NWIN, WBITS, slot, nullptr, MODE_TABLES);  // fixed-base tables: one bucket set per vector
  if (w_begin == 0 && w_end == NWIN && window_sums_dev == nullptr && nvec >= 1 && ctx->endo && n > 0 && nvec * nwin_of(16, true) <= MAXLW) {
    const int wbits = pick_window_bits(ctx, n, nvec, true);  // endomorphism: half-length scalars over 2n points
    return launch_impl(ctx, scalars_dev, n, nvec, 0, nwin_of(wbits, true), wbits, slot, nullptr, MODE_HALVES);
  }
  if (w_begin == 0 && w_end == NWIN && window_sums_dev == nullptr && nvec >= 1 && nvec * NWIN <= MAXLW) {
    const int wbits = pick_window_bits(ctx, n, nvec);
    return launch_impl(ctx, scalars_dev, n, nvec, 0, nwin_of(wbits), wbits, slot, nullptr);
  }
  return launch_impl(ctx, scalars_dev, n, nvec, w_begin, w_end, WBITS, slot, window_sums_dev);
}

int msm_hip_launch_half_windows_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int hw_begin, int hw_end,
                                                   int slot, void* window_sums_dev) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  if (!ctx->endo && ctx->n_bases) return MSM_HIP_ERR_INVALID_ARG;  // needs bases set with MSM_HIP_BASES_ENDOMORPHISM
  return launch_impl(ctx, scalars_dev, n, nvec, hw_begin, hw_end, WBITS, slot, window_sums_dev, MODE_HALVES);
}

int msm_hip_launch_vwindows_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int v_begin, int v_end, int slot,
                                         void* sums_dev) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  if (!ctx->wide_bits) return ctx->n_bases || !n ? MSM_HIP_ERR_INVALID_ARG : MSM_HIP_ERR_NO_BASES;  // needs bases set with MSM_HIP_BASES_PRECOMPUTE_WIDE
  if (v_begin < 0 || v_end <= v_begin) return MSM_HIP_ERR_INVALID_ARG;
  return launch_impl(ctx, scalars_dev, n, nvec, 0, wide_tables_of(ctx->wide_bits), ctx->wide_bits, slot, sums_dev, MODE_WIDE, v_begin, v_end - v_begin);
}

int msm_hip_launch_windows_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end, int slot,
                                        void* window_sums_dev) {
  return msm_hip_launch_windows_batch_device(ctx, scalars_dev, n, 1, w_begin, w_end, slot, window_sums_dev);
}

int msm_hip_launch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int slot) {
  return msm_hip_launch_windows_device(ctx, scalars_dev, n, 0, NWIN, slot, nullptr);
}

int msm_hip_slot_wait_stream(msm_hip_ctx* ctx, int slot, void* foreign_stream) {
  if (!ctx || slot < 0 || slot >= NSLOT) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipStreamWaitEvent(static_cast<hipStream_t>(foreign_stream), ctx->slot[slot].done, 0));
  return MSM_HIP_OK;
}

int msm_hip_slot_sync(msm_hip_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot >= NSLOT) return MSM_HIP_ERR_INVALID_ARG;
  Slot& s = ctx->slot[slot];
  if (!s.pending) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  return wait_slot(ctx, s);
}

int msm_hip_finish_batch(msm_hip_ctx* ctx, int slot, uint8_t* out_xyz) {
  if (!ctx || !out_xyz || slot < 0 || slot >= NSLOT) return MSM_HIP_ERR_INVALID_ARG;
  Slot& s = ctx->slot[slot];
  // fixed-base launches leave ONE sum per vector (every table already carries its power of two): nothing to combine but the copy
  const int nwin = s.merged ? 1 : s.wide_bits ? wide_vwin_of(s.wide_bits) : nwin_of(s.wbits, s.halves);
  if (!s.pending || !s.to_host || s.pairs || s.w_count != (s.wide_bits ? wide_tables_of(s.wide_bits) : nwin_of(s.wbits, s.halves))) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  int rc = wait_slot(ctx, s);
  if (rc) return rc;
  auto t0 = std::chrono::steady_clock::now();
  std::atomic<bool> all_ok{true};
  if (s.parts) {  // one MSM (wide tables: up to 12), its windows as bit-plane sums: the windows' positional sums side by side, then the chain over them
    uint8_t sums[24 * MAX_JB];
    const size_t jb = ctx->jb;
    const int total = s.wide_bits ? s.nvec * nwin : nwin;
    combine_pool().run(total, [&](int w) {
      if (!ctx->ops->window_from_planes(s.h_wsums + (size_t)w * PLANES_PER_WINDOW * jb, sums + jb * (size_t)w)) all_ok = false;
    });
    if (s.wide_bits) {  // virtual windows: sum_hi W_hi + 2^15 sum_hi hi TC_hi (host_g1.h), one chain per MSM of the launch
      combine_pool().run(s.nvec, [&](int v) {
        if (!ctx->ops->combine_wide(sums + jb * (size_t)v * nwin, s.h_wsums + (size_t)v * nwin * PLANES_PER_WINDOW * jb, nwin, out_xyz + jb * (size_t)v)) all_ok = false;
      });
    } else if (!ctx->ops->combine_windows(sums, nwin, s.wbits, out_xyz)) all_ok = false;
  } else {
    combine_pool().run(s.nvec, [&](int v) {  // one independent Horner chain per MSM of the launch: side by side when there are several
      if (!ctx->ops->combine_windows(s.h_wsums + (size_t)v * nwin * ctx->jb, nwin, s.wbits, out_xyz + ctx->jb * (size_t)v)) all_ok = false;
    });
  }
  if (!all_ok) return MSM_HIP_ERR_HIP;
  ctx->stage_ms[8] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MSM_HIP_OK;
}

int msm_hip_finish(msm_hip_ctx* ctx, int slot, uint8_t out_xyz[96]) {
  if (!ctx || slot < 0 || slot >= NSLOT || ctx->slot[slot].nvec != 1) return MSM_HIP_ERR_INVALID_ARG;
  return msm_hip_finish_batch(ctx, slot, out_xyz);
}

int msm_hip_run_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, uint8_t out_xyz[96]) {
  if (!out_xyz || !ctx) return MSM_HIP_ERR_INVALID_ARG;
  ctx->sync_call = true;
  int rc = msm_hip_launch_device(ctx, scalars_dev, n, 0);
  ctx->sync_call = false;
  if (rc) return rc;
  return msm_hip_finish(ctx, 0, out_xyz);
}

}  // extern "C"

namespace {
// host scalars -> the slot's own staging buffer (copy stream) -> windows [w_begin, w_end) of one MSM into `slot`
// (windows in units of the reference's 16-bit windows; `auto_bits`: a whole MSM whose window size follows n)
int launch_host_windows(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, int w_begin, int w_end, int slot, void* window_sums_dev,
                        bool auto_bits = false) {
  int rc = check_run_args(ctx, scalars_host, n);
  if (rc) return rc;
  if (slot < 0 || slot >= NSLOT) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  Slot& s = ctx->slot[slot];
  if (s.pending) return MSM_HIP_ERR_SLOT_BUSY;
  if (n == 0) return launch_impl(ctx, scalars_host, 0, 1, w_begin, w_end, WBITS, slot, window_sums_dev);
  if ((rc = setup_slot(ctx, s))) return rc;
  if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (n > s.cap_host_scalars) {  // the slot's own staging buffer: idle, since the slot is not pending
    s.cap_host_scalars = 0;
    if ((rc = dev_alloc(ctx, s.d_host_scalars, n * 8))) return rc;
    s.cap_host_scalars = n;
  }
  // H2D on the copy stream, so that it overlaps whatever the main and reduce streams still hold of earlier launches
  // (a caller that alternates two slots gets the copy of MSM i+1 under the device work of MSM i); the main stream waits
  // for it on the device.  From pageable memory the call returns when the bytes have left the caller's buffer; from
  // pinned memory (hipHostMalloc / hipHostRegister) at once -- the buffer must then stay untouched until finish / slot_sync.
  HIP_TRY(ctx, hipMemcpyAsync(s.d_host_scalars, scalars_host, n * 32, hipMemcpyHostToDevice, ctx->copy_stream));
  HIP_TRY(ctx, hipEventRecord(s.staged, ctx->copy_stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.staged, 0));
  if (auto_bits) return msm_hip_launch_windows_batch_device(ctx, s.d_host_scalars, n, 1, 0, NWIN, slot, nullptr);
  return launch_impl(ctx, s.d_host_scalars, n, 1, w_begin, w_end, WBITS, slot, window_sums_dev);
}
}  // namespace

extern "C" {

int msm_hip_launch(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, int slot) {
  return launch_host_windows(ctx, scalars_host, n, 0, NWIN, slot, nullptr, true);
}

int msm_hip_run(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]) {
  if (!out_xyz || !ctx) return MSM_HIP_ERR_INVALID_ARG;
  // Parts (round 5): 32 n bytes over the host link come before anything can run.  From 2^19 points on the call is the sum of sub-MSMs over
  // ranges of the points, one result slot each: part k + 1's scalars arrive while part k is sorted and accumulated, and the results are added
  // on the host.  (Not with fixed-base tables: their launches are shaped by the table count.)
  int parts = ctx->precomputed || ctx->wide_bits || !scalars_host ? 1 : upload_parts(n, NSLOT, 20, 22);  // (2^20: 2.44 -> 2.38 ms, 2^22: 8.61 -> 6.91; 2^19: slower)
  for (int k = 0; k < parts; k++)
    if (ctx->slot[k].pending) parts = 1;  // the caller has launches of its own in flight: the plain path (which reports a busy slot 0)
  if (parts > 1 && n <= ctx->n_bases) {
    uint8_t sums[NSLOT * MAX_JB];
    int rc = MSM_HIP_OK, launched = 0;
    for (int k = 0; k < parts && !rc; k++) {
      const size_t first = n / parts * k + (n % parts < (size_t)k ? n % parts : (size_t)k), next = n / parts * (k + 1) + (n % parts < (size_t)(k + 1) ? n % parts : (size_t)(k + 1));
      ctx->launch_base_off = first;
      rc = msm_hip_launch(ctx, scalars_host + first * 32, next - first, k);
      ctx->launch_base_off = 0;
      if (!rc) launched++;
    }
    for (int k = 0; k < launched; k++) {
      const int frc = msm_hip_finish(ctx, k, sums + (size_t)k * ctx->jb);
      if (!rc) rc = frc;
    }
    if (!rc && !ctx->ops->combine_windows(sums, parts, 0, out_xyz)) rc = MSM_HIP_ERR_NONCANONICAL;  // (window_bits = 0: the plain sum)
    return rc;
  }
  ctx->sync_call = true;
  int rc = msm_hip_launch(ctx, scalars_host, n, 0);
  ctx->sync_call = false;
  if (rc) return rc;
  return msm_hip_finish(ctx, 0, out_xyz);
}

int msm_hip_run_batch_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, size_t batch, uint8_t* out_xyz) {
  if (!out_xyz && batch) return MSM_HIP_ERR_INVALID_ARG;
  int rc = check_run_args(ctx, scalars_dev, n);
  if (rc) return rc;
  const uint8_t* sc = static_cast<const uint8_t*>(scalars_dev);
  return run_batch_groups(ctx, n, batch, out_xyz, [&](size_t, size_t first, size_t, const void** dev) {
    *dev = sc + first * n * 32;
    return (int)MSM_HIP_OK;
  });
}

int msm_hip_run_batch(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz) {
  if (!out_xyz && batch) return MSM_HIP_ERR_INVALID_ARG;
  int rc = check_run_args(ctx, scalars_host, n);
  if (rc) return rc;
  if (n == 0) {
    if (batch) memset(out_xyz, 0, ctx->jb * batch);
    return MSM_HIP_OK;
  }
  ON_DEVICE(ctx);
  const size_t vec = n * 32, entry = vec * batch_group(ctx, n, batch);
  if (entry * NSLOT > ctx->cap_batch_stage) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cap_batch_stage = 0;
    if ((rc = dev_alloc(ctx, ctx->d_batch_stage, entry * NSLOT))) return rc;
    ctx->cap_batch_stage = entry * NSLOT;
  }
  // group j is staged in ring entry j % NSLOT on the main stream just ahead of its own sort (the entry's previous
  // reader, group j - NSLOT, was finished DEPTH + 1 iterations ago)
  return run_batch_groups(ctx, n, batch, out_xyz, [&](size_t j, size_t first, size_t count, const void** dev) {
    uint8_t* stage = ctx->d_batch_stage + (j % NSLOT) * entry;
    if (hipMemcpyAsync(stage, scalars_host + first * vec, count * vec, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return (int)MSM_HIP_ERR_HIP;
    *dev = stage;
    return (int)MSM_HIP_OK;
  });
}

int msm_hip_run_windows_device(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end,
                                     void* window_sums_dev) {
  if (!window_sums_dev) return MSM_HIP_ERR_INVALID_ARG;
  int rc = msm_hip_launch_windows_device(ctx, scalars_dev, n, w_begin, w_end, 0, window_sums_dev);
  if (rc) return rc;
  return msm_hip_slot_sync(ctx, 0);
}

int msm_hip_combine_windows_bn254(const uint8_t* window_sums_host, int num_windows, uint8_t out_xyz[96]) {
  if (!window_sums_host || !out_xyz || num_windows < 1 || num_windows > NWIN) return MSM_HIP_ERR_INVALID_ARG;
  if (!bn254::host::combine_windows(window_sums_host, num_windows, WBITS, out_xyz)) return MSM_HIP_ERR_NONCANONICAL;
  return MSM_HIP_OK;
}

int msm_hip_combine_windows_batch_curve(int curve, const uint8_t* window_sums_host, int num_windows, int nvec, uint8_t* out_xyz) {
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  if (!window_sums_host || !out_xyz || num_windows < 1 || num_windows > NWIN || nvec < 0) return MSM_HIP_ERR_INVALID_ARG;
  const CurveOps* ops = curve_ops(curve);
  const size_t jb = 12 * (size_t)ops->coord_words;
  std::atomic<bool> ok{true};
  // independent Horner chains (47 us each on one core): side by side on the combine pool when there are several
  combine_pool().run(nvec, [&](int v) {
    if (!ops->combine_windows(window_sums_host + (size_t)v * num_windows * jb, num_windows, WBITS, out_xyz + (size_t)v * jb)) ok = false;
  });
  return ok ? MSM_HIP_OK : MSM_HIP_ERR_NONCANONICAL;
}

int msm_hip_combine_vwindows_batch_curve(int curve, const uint8_t* pairs_host, int num_vwindows, int nvec, uint8_t* out_xyz) {
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  if (!pairs_host || !out_xyz || num_vwindows < 1 || num_vwindows > 16 || nvec < 0) return MSM_HIP_ERR_INVALID_ARG;
  const CurveOps* ops = curve_ops(curve);
  const size_t jb = 12 * (size_t)ops->coord_words;
  std::atomic<bool> ok{true};
  combine_pool().run(nvec, [&](int v) {  // one short chain per MSM (2 additions per virtual window + 15 doublings), side by side
    if (!ops->combine_wide_pairs(pairs_host + (size_t)v * num_vwindows * 2 * jb, num_vwindows, out_xyz + (size_t)v * jb)) ok = false;
  });
  return ok ? MSM_HIP_OK : MSM_HIP_ERR_NONCANONICAL;
}

int msm_hip_g1_to_affine_bn254(const uint8_t xyz[96], uint8_t out_xy[64]) {
  if (!xyz || !out_xy) return MSM_HIP_ERR_INVALID_ARG;
  const int r = bn254::host::to_affine64(xyz, out_xy);
  return r < 0 ? MSM_HIP_ERR_NONCANONICAL : r;
}

int msm_hip_combine_windows_curve(int curve, const uint8_t* window_sums_host, int num_windows, uint8_t out_xyz[96]) {
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  if (!window_sums_host || !out_xyz || num_windows < 1 || num_windows > NWIN) return MSM_HIP_ERR_INVALID_ARG;
  if (!curve_ops(curve)->combine_windows(window_sums_host, num_windows, WBITS, out_xyz)) return MSM_HIP_ERR_NONCANONICAL;
  return MSM_HIP_OK;
}

int msm_hip_g1_to_affine_curve(int curve, const uint8_t xyz[96], uint8_t out_xy[64]) {
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  if (!xyz || !out_xy) return MSM_HIP_ERR_INVALID_ARG;
  const int r = curve_ops(curve)->to_affine64(xyz, out_xy);
  return r < 0 ? MSM_HIP_ERR_NONCANONICAL : r;
}

// The one-shot keeps ONE context per device alive between calls (the reference creates and drops its wgpu device on every call,
// src/cuzk/msm.rs:88-94; here that costs 0.3 ms of creation, ~1 ms of first-use allocations and 3.4 ms of hipFree per call --
// more than the MSM).  Process-wide, mutex-protected (one-shot calls on one device are serialised, as a context requires);
// msm_hip_oneshot_release() drops the kept contexts; MSM_HIP_ONESHOT_KEEP=0 restores create / destroy per call.
namespace {
constexpr int ONESHOT_MAX_DEVICES = 64;
constexpr int ONESHOT_MAX_PARTS = 4;  // contexts per (curve, device): a large one-shot MSM runs as up to this many sub-MSMs over ranges of the points (below)
std::mutex g_oneshot_mutex;
msm_hip_ctx* g_oneshot_ctx[MSM_HIP_NUM_CURVES][ONESHOT_MAX_DEVICES][ONESHOT_MAX_PARTS] = {};
inline bool oneshot_keep() {
  static const bool v = [] { const char* e = getenv("MSM_HIP_ONESHOT_KEEP"); return !(e && e[0] == '0'); }();
  return v;
}
}  // namespace

}  // extern "C"

namespace {
// The one-shot call on a kept (or fresh) context, OVERLAPPED (round 5): the reference uploads everything, then dispatches (src/cuzk/msm.rs:84-94,
// 441-480); here the 32 n bytes of scalars go first and their recode + sort (everything that needs the scalars alone: launch phase 1) runs while
// the 64 n bytes of points are still arriving -- in chunks on the copy stream, each converted to the device form (and its endomorphism image
// made) as soon as it has landed --, and only the SMVP (launch phase 2) waits for the last chunk.  Hidden: the sort (~0.26 ms at 2^20), the
// conversion kernels (~0.1 ms) and one host synchronisation.  MSM_HIP_ONESHOT_OVERLAP=0: upload, convert, then run (rounds 1 - 4).
// Two halves: oneshot_enqueue queues everything (copies on `cs` -- the copy stream of the call's FIRST part, so that the parts' uploads follow each
// other instead of sharing the link -- and both launch phases), oneshot_collect waits for the result.  `he_out`: the first HIP error of the
// chunk loop (the launch is completed regardless, its result discarded).
int oneshot_enqueue(msm_hip_ctx* ctx, hipStream_t cs, const uint8_t* xy_host, const uint8_t* scalars_host, size_t n, hipError_t* he_out, bool* queued) {
  ON_DEVICE(ctx);
  *he_out = hipSuccess;
  *queued = false;  // true: the launch was taken to its second phase -- oneshot_collect has to follow, whatever this function returns
  const uint32_t flags = resolve_base_flags(ctx, n, 0);
  const bool endo = (flags & MSM_HIP_BASES_ENDOMORPHISM) != 0;
  int rc = reserve_bases(ctx, n, flags);  // (waits for the main stream: nothing is reading the old bases)
  if (rc) return rc;
  Slot& s = ctx->slot[0];
  if (s.pending) return MSM_HIP_ERR_SLOT_BUSY;
  if ((rc = setup_slot(ctx, s))) return rc;
  if (!ctx->bases_ready) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->bases_ready, hipEventDisableTiming));
  if (n > s.cap_host_scalars) {
    s.cap_host_scalars = 0;
    if ((rc = dev_alloc(ctx, s.d_host_scalars, n * 8))) return rc;
    s.cap_host_scalars = n;
  }
  // 1. the scalars, and their sort
  HIP_TRY(ctx, hipMemcpyAsync(s.d_host_scalars, scalars_host, n * 32, hipMemcpyHostToDevice, cs));
  HIP_TRY(ctx, hipEventRecord(s.staged, cs));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.staged, 0));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_err, 0, 4, cs));  // (ahead of every chunk's copy, hence of every conversion)
  ctx->n_bases = n;  // (what the launch checks n against; the records themselves follow below)
  ctx->endo = endo;
  const LaunchMode mode = endo ? MODE_HALVES : MODE_PLAIN;
  const int wbits = pick_window_bits(ctx, n, 1, endo), w_end = nwin_of(wbits, endo);
  ctx->sync_call = true;
  rc = launch_impl(ctx, s.d_host_scalars, n, 1, 0, w_end, wbits, 0, nullptr, mode, 0, 0, 1);
  if (rc) {
    ctx->sync_call = false;
    ctx->n_bases = 0;
    return rc;
  }
  // 2. the points behind the scalars on the copy stream, in a few chunks (2^18 points = 16 MiB: smaller ones cost the pageable copy path more
  //    than they hide, profiles/r05_oneshot.txt); every chunk is converted on ANOTHER stream (the second reduce stream, idle in this call shape)
  //    as soon as it has landed, so the copies follow each other without waiting for kernels
  static const size_t chunk = [] {  // points per chunk (MSM_HIP_ONESHOT_CHUNK_LOG: tuning aid)
    const char* e = getenv("MSM_HIP_ONESHOT_CHUNK_LOG");
    const int l = e ? atoi(e) : 18;
    return (size_t)1 << (l >= 10 && l <= 28 ? l : 18);
  }();
  hipStream_t conv = ctx->reduce_stream[NREDUCE - 1];
  hipError_t he = hipSuccess;
  int k = 0;
  for (size_t first = 0; first < n && he == hipSuccess; first += chunk, k++) {
    const size_t count = n - first < chunk ? n - first : chunk;
    uint32_t* dst = ctx->d_bases + first * 2 * (size_t)ctx->ops->coord_words;
    hipEvent_t& landed = ctx->chunk_landed[k & 7];
    if (!landed && (he = hipEventCreateWithFlags(&landed, hipEventDisableTiming)) != hipSuccess) break;
    if ((he = hipMemcpyAsync(dst, xy_host + first * ctx->pb, count * ctx->pb, hipMemcpyHostToDevice, cs)) != hipSuccess) break;
    if ((he = hipEventRecord(landed, cs)) != hipSuccess) break;
    if ((he = hipStreamWaitEvent(conv, landed, 0)) != hipSuccess) break;
    hipLaunchKernelGGL(ctx->ops->convert_points, dim3(blocks_for(count, 256)), dim3(256), 0, conv, dst, dst, count, flags, ctx->d_err);
    if (endo) hipLaunchKernelGGL(ctx->ops->endo_points, dim3(blocks_for(count, 256)), dim3(256), 0, conv, ctx->d_bases, n, first, count);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipEventRecord(ctx->bases_ready, conv);
  if (he == hipSuccess) he = hipStreamWaitEvent(ctx->stream, ctx->bases_ready, 0);
  // 3. the rest of the launch (on failure above too: phase 1 left the slot's streams mid-launch -- the bases it then reads are whatever arrived,
  //    and the result is discarded)
  *queued = true;
  rc = launch_impl(ctx, s.d_host_scalars, n, 1, 0, w_end, wbits, 0, nullptr, mode, 0, 0, 2);
  ctx->sync_call = false;
  *he_out = he;
  return rc;
}

// ... and the wait: the result of the part (discarded after a failure of its enqueue half), then the conversion's error word
int oneshot_collect(msm_hip_ctx* ctx, int rc, hipError_t he, uint8_t* out_xyz) {
  ON_DEVICE(ctx);
  hipStream_t conv = ctx->reduce_stream[NREDUCE - 1];
  uint8_t scratch[MAX_JB];
  const int frc = rc ? rc : msm_hip_finish(ctx, 0, he == hipSuccess ? out_xyz : scratch);
  uint32_t base_bits = 0;  // the conversion's error word (read last: the launch was queued behind the copy stream before the host waits for anything)
  if (he == hipSuccess) he = hipStreamSynchronize(conv);
  if (he == hipSuccess) he = hipMemcpy(&base_bits, ctx->d_err, 4, hipMemcpyDeviceToHost);
  if (he != hipSuccess) {
    ctx->last_hip_error = (int)he;
    ctx->n_bases = 0;
    return MSM_HIP_ERR_HIP;
  }
  if (const int brc = err_from_bits(base_bits)) {  // a non-canonical coordinate: the bases are not usable (as msm_hip_set_bases_* reports it)
    ctx->n_bases = 0;
    return brc;
  }
  return frc;
}
}  // namespace

extern "C" {

int msm_hip_msm_curve(int curve, const uint8_t* xy_host, const uint8_t* scalars_host, size_t n, uint8_t* out_xyz) {
  int dev = 0;
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  if (hipGetDevice(&dev) != hipSuccess) return MSM_HIP_ERR_NO_DEVICE;  // the caller's current device (0 unless it chose another)
  if (!out_xyz || ((!xy_host || !scalars_host) && n)) return MSM_HIP_ERR_INVALID_ARG;
  const bool keep = oneshot_keep() && dev >= 0 && dev < ONESHOT_MAX_DEVICES;
  std::lock_guard<std::mutex> lock(g_oneshot_mutex);  // (also without `keep`: the part policy below is process-wide state)
  static const bool overlap = [] { const char* e = getenv("MSM_HIP_ONESHOT_OVERLAP"); return !e || atoi(e) != 0; }();
  // Parts (round 5): a large one-shot MSM is upload-bound -- 96 n bytes over the host link against ~1 ms of SMVP at 2^20 --, and the SMVP cannot start
  // before the last point has arrived.  Sum over the points is sum over RANGES of the points: the call runs as `parts` sub-MSMs, each on a context
  // of its own, their uploads queued one behind the other on one copy stream, so that part k accumulates its buckets while part k + 1 is still
  // arriving and only the LAST part's SMVP (n / parts points) follows the upload.  The results are added on the host (parts - 1 additions).
  // 2 parts from 2^19 points on (MSM_HIP_ONESHOT_PARTS: tuning aid; more parts pay more per-part sorting, stitching and bucket reducing).
  const bool overlapped = overlap && n > 0 && n <= MAX_POINTS / 2;
  const int parts = overlapped ? upload_parts(n, ONESHOT_MAX_PARTS, 19, 30) : 1;  // (2^19: 2.00 -> 1.93 ms, 2^20: 3.55 -> 3.08, 2^22: 13.2 -> 11.1; three parts never better)
  msm_hip_ctx* ctxs[ONESHOT_MAX_PARTS] = {};
  int rc = MSM_HIP_OK;
  for (int k = 0; k < parts && !rc; k++) {
    if (keep) ctxs[k] = g_oneshot_ctx[curve][dev][k];
    if (!ctxs[k]) {
      rc = msm_hip_ctx_create_curve(&ctxs[k], dev, curve);
      if (!rc && keep) g_oneshot_ctx[curve][dev][k] = ctxs[k];
    }
  }
  if (!rc && !overlapped) {
    rc = msm_hip_set_bases(ctxs[0], xy_host, n, 0);
    if (!rc) rc = msm_hip_run(ctxs[0], scalars_host, n, out_xyz);
  } else if (!rc) {
    msm_hip_ctx* c0 = ctxs[0];
    if (!c0->copy_stream) {
      DeviceGuard guard(c0->device);
      if (hipStreamCreateWithFlags(&c0->copy_stream, hipStreamNonBlocking) != hipSuccess) rc = MSM_HIP_ERR_HIP;
    }
    int prc[ONESHOT_MAX_PARTS] = {};
    hipError_t phe[ONESHOT_MAX_PARTS] = {};
    bool queued[ONESHOT_MAX_PARTS] = {};
    uint8_t sums[ONESHOT_MAX_PARTS * MAX_JB];
    const size_t pb = c0->pb, jb = c0->jb;
    size_t first[ONESHOT_MAX_PARTS + 1];
    for (int k = 0; k <= parts; k++) first[k] = n / parts * k + (n % parts < (size_t)k ? n % parts : (size_t)k);
    for (int k = 0; k < parts && !rc; k++) {
      prc[k] = oneshot_enqueue(ctxs[k], c0->copy_stream, xy_host + first[k] * pb, scalars_host + first[k] * 32, first[k + 1] - first[k], &phe[k], &queued[k]);
      if (prc[k] && !queued[k]) rc = prc[k];  // nothing of this part is in flight: stop queueing, drain the earlier ones
    }
    for (int k = 0; k < parts; k++) {
      if (!queued[k]) continue;
      const int crc = oneshot_collect(ctxs[k], prc[k], phe[k], parts == 1 ? out_xyz : sums + (size_t)k * jb);
      if (!rc) rc = crc;
    }
    if (!rc && parts > 1 && !c0->ops->combine_windows(sums, parts, 0, out_xyz)) rc = MSM_HIP_ERR_NONCANONICAL;  // (window_bits = 0: the plain sum of the records)
  }
  if (!keep)
    for (msm_hip_ctx* c : ctxs)
      if (c) msm_hip_ctx_destroy(c);
  return rc;
}

int msm_hip_msm_bn254_g1(const uint8_t* xy_host, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]) {
  return msm_hip_msm_curve(MSM_HIP_CURVE_BN254_G1, xy_host, scalars_host, n, out_xyz);
}

// test hook: the one-shot call's part policy (0, 0 restores the default: 2 parts from 2^19 points on) -- lets a test run several parts on small inputs
int msm_hip_test_oneshot_parts(int parts, size_t min_points) {
  if (parts < 0 || parts > ONESHOT_MAX_PARTS) return MSM_HIP_ERR_INVALID_ARG;
  g_oneshot_parts.store(parts);
  g_oneshot_parts_min_n.store(min_points);
  return MSM_HIP_OK;
}

void msm_hip_oneshot_release(void) {
  std::lock_guard<std::mutex> lock(g_oneshot_mutex);
  for (auto& per_curve : g_oneshot_ctx)
    for (auto& per_device : per_curve)
      for (msm_hip_ctx*& c : per_device) {
        if (c) msm_hip_ctx_destroy(c);
        c = nullptr;
      }
}

int msm_hip_sample_scalars_device(msm_hip_ctx* ctx, uint64_t seed, size_t n, void* scalars_dev) {
  if (!ctx || (!scalars_dev && n)) return MSM_HIP_ERR_INVALID_ARG;
  if (n == 0) return MSM_HIP_OK;
  ON_DEVICE(ctx);
  hipLaunchKernelGGL(ctx->ops->sample_scalars, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, seed, n, static_cast<uint32_t*>(scalars_dev));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM_HIP_OK;
}

int msm_hip_sample_points_device(msm_hip_ctx* ctx, uint64_t seed, size_t n, void* xy_dev) {
  if (!ctx || (!xy_dev && n)) return MSM_HIP_ERR_INVALID_ARG;
  if (n == 0) return MSM_HIP_OK;
  if (!ctx->ops->sample_points) return MSM_HIP_ERR_INVALID_ARG;  // (no device sampler for this curve: none at present)
  ON_DEVICE(ctx);
  hipLaunchKernelGGL(ctx->ops->sample_points, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, seed, n, static_cast<uint32_t*>(xy_dev));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM_HIP_OK;
}

int msm_hip_last_stage_ms(msm_hip_ctx* ctx, float* ms, int cap) {
  if (!ctx || !ms) return MSM_HIP_ERR_INVALID_ARG;
  int k = cap < 9 ? cap : 9;
  for (int i = 0; i < k; i++) ms[i] = ctx->stage_ms[i];
  return k;
}

// ---- stage read-back ------------------------------------------------------------------------------------------------
static int read_back(msm_hip_ctx* ctx, void* out, const void* src, size_t bytes, size_t cap_bytes) {
  if (!ctx || !out) return MSM_HIP_ERR_INVALID_ARG;
  if (bytes > cap_bytes) return MSM_HIP_ERR_INVALID_ARG;
  if (bytes == 0) return MSM_HIP_OK;
  ON_DEVICE(ctx);
  for (hipStream_t r : ctx->reduce_stream) HIP_TRY(ctx, hipStreamSynchronize(r));
  HIP_TRY(ctx, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM_HIP_OK;
}

int msm_hip_set_stage_timing(msm_hip_ctx* ctx, int level) {
  if (!ctx || level < 0 || level > 2) return MSM_HIP_ERR_INVALID_ARG;
  ctx->timing_level = level;
  return MSM_HIP_OK;
}

int msm_hip_set_scalar_format(msm_hip_ctx* ctx, uint32_t format) {
  if (!ctx || (format != MSM_HIP_SCALARS_CANONICAL && format != MSM_HIP_SCALARS_MONT256)) return MSM_HIP_ERR_INVALID_ARG;
  if (format == MSM_HIP_SCALARS_MONT256 && !ctx->ops->scalars_from_mont256) return MSM_HIP_ERR_INVALID_ARG;
  ctx->scalar_format = format;
  return MSM_HIP_OK;
}

int msm_hip_set_window_bits(msm_hip_ctx* ctx, int bits) {
  if (!ctx || (bits != 0 && bits != 12 && bits != 14 && bits != 16)) return MSM_HIP_ERR_INVALID_ARG;
  ctx->window_bits = bits;
  return MSM_HIP_OK;
}

int msm_hip_set_wide_bits(msm_hip_ctx* ctx, int bits) {
  if (!ctx || (bits != 0 && (bits < 16 || bits > 20))) return MSM_HIP_ERR_INVALID_ARG;
  ctx->wide_bits_choice = bits;
  return MSM_HIP_OK;
}

int msm_hip_wide_bits(const msm_hip_ctx* ctx) { return ctx ? ctx->wide_bits : MSM_HIP_ERR_INVALID_ARG; }

int msm_hip_wide_config(int curve, int bits, size_t n, int* digit_bits, int* tables, int* virtual_windows, int* top_shift) {
  if (curve < MSM_HIP_CURVE_BN254_G1 || curve > MSM_HIP_CURVE_BLS12_381_G2 || (bits != 0 && (bits < 16 || bits > 20))) return MSM_HIP_ERR_INVALID_ARG;
  msm_hip_ctx probe;  // (only the two fields the policy reads)
  probe.curve = curve;
  probe.wide_bits_choice = bits;
  const int wb = pick_wide_bits(&probe, n);
  if (wb < 0) return MSM_HIP_ERR_INVALID_ARG;
  if (digit_bits) *digit_bits = wb;
  if (tables) *tables = wide_tables_of(wb);
  if (virtual_windows) *virtual_windows = wide_vwin_of(wb);
  if (top_shift) *top_shift = wide_top_shift(curve, wb);
  return MSM_HIP_OK;
}

int msm_hip_window_config(int bits, int* num_windows, int* buckets_per_window) {
  if (bits != 12 && bits != 14 && bits != 16) return MSM_HIP_ERR_INVALID_ARG;
  if (num_windows) *num_windows = nwin_of(bits);
  if (buckets_per_window) *buckets_per_window = 1 << (bits - 1);
  return MSM_HIP_OK;
}

int msm_hip_last_window_bits(msm_hip_ctx* ctx) { return ctx ? ctx->last_wbits : MSM_HIP_ERR_INVALID_ARG; }

int msm_hip_endomorphism_window_count(int bits) {
  if (bits != 12 && bits != 14 && bits != 16) return MSM_HIP_ERR_INVALID_ARG;
  return nwin_of(bits, true);
}

int msm_hip_uses_endomorphism(const msm_hip_ctx* ctx) { return ctx ? (ctx->endo ? 1 : 0) : MSM_HIP_ERR_INVALID_ARG; }

int msm_hip_batch_group_size(msm_hip_ctx* ctx, size_t n) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  return (int)batch_group(ctx, n, (size_t)MAXLW);
}

int msm_hip_set_fine_hist_min_n(msm_hip_ctx* ctx, size_t n) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  ctx->fine_hist_min_n = n;
  return MSM_HIP_OK;
}

int msm_hip_set_debug(msm_hip_ctx* ctx, int keep_digit_planes) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  ctx->debug = keep_digit_planes != 0;
  return MSM_HIP_OK;
}

int msm_hip_read_digits(msm_hip_ctx* ctx, uint16_t* out, size_t cap_elems) {
  if (!ctx || !ctx->last_has_digits) return MSM_HIP_ERR_INVALID_ARG;
  return read_back(ctx, out, ctx->d_digits, ctx->last_n * ctx->last_w_count * 2, cap_elems * 2);
}
int msm_hip_read_col_ptr(msm_hip_ctx* ctx, uint32_t* out, size_t cap_elems) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  const size_t half = (size_t)1 << (ctx->last_wbits - 1);  // [w][half + 1]
  return read_back(ctx, out, ctx->slot[ctx->last_slot].d_col_ptr, (size_t)ctx->last_w_count * (half + 1) * 4, cap_elems * 4);
}
int msm_hip_read_val_idxs(msm_hip_ctx* ctx, uint32_t* out, size_t cap_elems) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  // out[w][n]; only the first col_ptr[w][32768] entries of each window are meaningful
  const size_t n = ctx->last_n;
  if (!out || n * ctx->last_w_count > cap_elems) return MSM_HIP_ERR_INVALID_ARG;
  if (n == 0) return MSM_HIP_OK;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipMemcpy2DAsync(out, n * 4, ctx->d_val, ctx->last_stride * 4, n * 4, (size_t)ctx->last_w_count, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM_HIP_OK;
}

int msm_hip_read_buckets(msm_hip_ctx* ctx, uint8_t* out, size_t cap_bytes) {
  if (!ctx || !out) return MSM_HIP_ERR_INVALID_ARG;
  const size_t count = (size_t)ctx->last_w_count << (ctx->last_wbits - 1);  // [w][2^(bits-1)]
  if (count * ctx->jb > cap_bytes) return MSM_HIP_ERR_INVALID_ARG;
  if (count == 0) return MSM_HIP_OK;
  ON_DEVICE(ctx);
  for (hipStream_t r : ctx->reduce_stream) HIP_TRY(ctx, hipStreamSynchronize(r));
  int rc = ensure_stage(ctx, count * ctx->jb);
  if (rc) return rc;
  hipLaunchKernelGGL(ctx->ops->export_buckets, dim3(blocks_for(count, 256)), dim3(256), 0, ctx->stream, ctx->slot[ctx->last_slot].d_buckets,
                     reinterpret_cast<uint32_t*>(ctx->d_stage), count);
  HIP_TRY(ctx, hipGetLastError());
  return read_back(ctx, out, ctx->d_stage, count * ctx->jb, cap_bytes);
}

int msm_hip_read_window_sums(msm_hip_ctx* ctx, uint8_t* out, size_t cap_bytes) {
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  const Slot& s = ctx->slot[ctx->last_slot];
  if (!s.parts) return read_back(ctx, out, s.d_wsums, (size_t)ctx->last_w_count * ctx->jb, cap_bytes);
  // the last launch handed its window sums to the host as bit-plane sums (k_bpr_planes): finish them here
  const size_t w = (size_t)ctx->last_w_count;
  if (!out || w * ctx->jb > cap_bytes || w > 24) return MSM_HIP_ERR_INVALID_ARG;
  const size_t plane_bytes = PLANES_PER_WINDOW * ctx->jb;
  {  // (the plane sums were written into the slot's pinned buffer by the launch's last kernel: complete once its reduce stream has drained)
    ON_DEVICE(ctx);
    for (hipStream_t r : ctx->reduce_stream) HIP_TRY(ctx, hipStreamSynchronize(r));
  }
  const uint8_t* planes = s.h_wsums;
  bool ok = true;
  for (size_t k = 0; k < w; k++) ok &= ctx->ops->window_from_planes(planes + k * plane_bytes, out + k * ctx->jb);
  return ok ? MSM_HIP_OK : MSM_HIP_ERR_NONCANONICAL;
}

// ---- op hooks -------------------------------------------------------------------------------------------------------
static int run_hook(msm_hip_ctx* ctx, const uint8_t* a, size_t a_bytes, const uint8_t* b, size_t b_bytes, uint8_t* out,
                    size_t out_bytes, uint8_t*& da, uint8_t*& db, uint8_t*& dout) {
  if (!ctx || !a || !out) return MSM_HIP_ERR_INVALID_ARG;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  int rc = ensure_stage(ctx, up16(a_bytes) + up16(b_bytes) + up16(out_bytes) + 16);
  if (rc) return rc;
  da = ctx->d_stage;
  db = da + up16(a_bytes);
  dout = db + up16(b_bytes);
  HIP_TRY(ctx, hipMemcpyAsync(da, a, a_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (b && b_bytes) HIP_TRY(ctx, hipMemcpyAsync(db, b, b_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (!b) db = nullptr;
  return MSM_HIP_OK;
}

int msm_hip_test_fq_op(msm_hip_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  if (n == 0) return MSM_HIP_OK;
  if (op < 0 || op > 9) return MSM_HIP_ERR_INVALID_ARG;
  uint8_t *da, *db, *dout;
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  const size_t cb = ctx->cb;
  int rc = run_hook(ctx, a, n * cb, b, b ? n * cb : 0, out, n * cb, da, db, dout);
  if (rc) return rc;
  hipLaunchKernelGGL(ctx->ops->test_fq, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, op, (const uint32_t*)da, (const uint32_t*)db,
                     (uint32_t*)dout, n);
  HIP_TRY(ctx, hipGetLastError());
  return read_back(ctx, out, dout, n * cb, n * cb);
}

int msm_hip_test_g1_op(msm_hip_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  if (n == 0) return MSM_HIP_OK;
  if (op < 0 || op > 4 || (op != 1 && !b)) return MSM_HIP_ERR_INVALID_ARG;
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  const size_t jb = ctx->jb;
  const size_t b_bytes = op == 0 ? n * jb : (op >= 2 ? n * ctx->pb : 0);
  uint8_t *da, *db, *dout;
  int rc = run_hook(ctx, a, n * jb, op == 1 ? nullptr : b, b_bytes, out, n * jb, da, db, dout);
  if (rc) return rc;
  hipLaunchKernelGGL(ctx->ops->test_g1, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, op, (const uint32_t*)da, (const uint32_t*)db,
                     (uint32_t*)dout, n);
  HIP_TRY(ctx, hipGetLastError());
  return read_back(ctx, out, dout, n * jb, n * jb);
}

int msm_hip_test_g1_mul_u32(msm_hip_ctx* ctx, const uint8_t* a, const uint32_t* k, uint8_t* out, size_t n) {
  if (n == 0) return MSM_HIP_OK;
  if (!k) return MSM_HIP_ERR_INVALID_ARG;
  uint8_t *da, *db, *dout;
  if (!ctx) return MSM_HIP_ERR_INVALID_ARG;
  const size_t jb = ctx->jb;
  int rc = run_hook(ctx, a, n * jb, reinterpret_cast<const uint8_t*>(k), n * 4, out, n * jb, da, db, dout);
  if (rc) return rc;
  hipLaunchKernelGGL(ctx->ops->test_g1_mul_u32, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)da,
                     (const uint32_t*)db, (uint32_t*)dout, n);
  HIP_TRY(ctx, hipGetLastError());
  return read_back(ctx, out, dout, n * jb, n * jb);
}

}  // extern "C"

#include "msm_mgpu.h"

// ---- the `_bn254` names of rounds 1 - 4: aliases of the curve-neutral entry points (include/msm_hip.h, last block)
extern "C" {
int msm_hip_set_bases_bn254(msm_hip_ctx* ctx, const uint8_t* xy_host, size_t n, uint32_t flags) { return msm_hip_set_bases(ctx, xy_host, n, flags); }
int msm_hip_set_bases_device_bn254(msm_hip_ctx* ctx, const void* xy_dev, size_t n, uint32_t flags) { return msm_hip_set_bases_device(ctx, xy_dev, n, flags); }
int msm_hip_run_bn254(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]) { return msm_hip_run(ctx, scalars_host, n, out_xyz); }
int msm_hip_run_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, uint8_t out_xyz[96]) { return msm_hip_run_device(ctx, scalars_dev, n, out_xyz); }
int msm_hip_launch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int slot) { return msm_hip_launch_device(ctx, scalars_dev, n, slot); }
int msm_hip_finish_bn254(msm_hip_ctx* ctx, int slot, uint8_t out_xyz[96]) { return msm_hip_finish(ctx, slot, out_xyz); }
int msm_hip_launch_bn254(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, int slot) { return msm_hip_launch(ctx, scalars_host, n, slot); }
int msm_hip_run_batch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, size_t batch, uint8_t* out_xyz) {
  return msm_hip_run_batch_device(ctx, scalars_dev, n, batch, out_xyz);
}
int msm_hip_run_batch_bn254(msm_hip_ctx* ctx, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz) {
  return msm_hip_run_batch(ctx, scalars_host, n, batch, out_xyz);
}
int msm_hip_run_windows_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end, void* window_sums_dev) {
  return msm_hip_run_windows_device(ctx, scalars_dev, n, w_begin, w_end, window_sums_dev);
}
int msm_hip_launch_windows_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int w_begin, int w_end, int slot, void* window_sums_dev) {
  return msm_hip_launch_windows_device(ctx, scalars_dev, n, w_begin, w_end, slot, window_sums_dev);
}
int msm_hip_launch_windows_batch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int w_begin, int w_end, int slot,
                                              void* window_sums_dev) {
  return msm_hip_launch_windows_batch_device(ctx, scalars_dev, n, nvec, w_begin, w_end, slot, window_sums_dev);
}
int msm_hip_finish_batch_bn254(msm_hip_ctx* ctx, int slot, uint8_t* out_xyz) { return msm_hip_finish_batch(ctx, slot, out_xyz); }
int msm_hip_launch_half_windows_batch_device_bn254(msm_hip_ctx* ctx, const void* scalars_dev, size_t n, int nvec, int hw_begin, int hw_end, int slot,
                                                   void* window_sums_dev) {
  return msm_hip_launch_half_windows_batch_device(ctx, scalars_dev, n, nvec, hw_begin, hw_end, slot, window_sums_dev);
}
int msm_hip_mgpu_set_bases_bn254(msm_hip_mgpu* m, const uint8_t* xy_host, size_t n, uint32_t flags) { return msm_hip_mgpu_set_bases(m, xy_host, n, flags); }
int msm_hip_mgpu_run_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]) { return msm_hip_mgpu_run(m, scalars_host, n, out_xyz); }
int msm_hip_mgpu_launch_batch_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, int nvec, int slot) {
  return msm_hip_mgpu_launch_batch(m, scalars_host, n, nvec, slot);
}
int msm_hip_mgpu_launch_batch_device_bn254(msm_hip_mgpu* m, const void* const* scalars_dev, size_t n, int nvec, int slot) {
  return msm_hip_mgpu_launch_batch_device(m, scalars_dev, n, nvec, slot);
}
int msm_hip_mgpu_finish_batch_bn254(msm_hip_mgpu* m, int slot, uint8_t* out_xyz) { return msm_hip_mgpu_finish_batch(m, slot, out_xyz); }
int msm_hip_mgpu_run_batch_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz) {
  return msm_hip_mgpu_run_batch(m, scalars_host, n, batch, out_xyz);
}
}  // extern "C"
