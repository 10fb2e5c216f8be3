// C++ host-side mirror of the reference's Rust surface for the MSM path, over the C ABI of msm_hip.h.
//
// The reference is a Rust crate (/root/reference/src/lib.rs); this image has no Rust toolchain, so the host side above
// the ABI is C++ (header-only, links libmsm_hip.so).  Names, argument meaning and error behaviour follow the reference:
//
//   sample_scalars / sample_points            src/lib.rs:20-42    (seeded here; thread_rng there)
//   scalars_to_bytes / points_to_bytes        src/lib.rs:50-65
//   cpu_msm                                   src/lib.rs:45-47    NOT provided: the product has no CPU path (the oracle is
//                                                                 test infrastructure)
//   run_webgpu_msm / compute_msm              src/lib.rs:76-82, src/cuzk/msm.rs:75-417
//   G1 == G1 (projective equality)            src/lib.rs:166
//
// Where the reference panics (no device gpu.rs:22,51; infinity in the input lib.rs:58; non-canonical bytes utils.rs:20;
// off-curve read-back msm.rs:399) these functions throw msm_webgpu::Error carrying the C-ABI code.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "msm_hip.h"

namespace msm_webgpu {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* where) : std::runtime_error(std::string(where) + ": " + msm_hip_strerror(c)), code(c) {}
};
inline void check(int code, const char* where) {
  if (code != MSM_HIP_OK) throw Error(code, where);
}

// field elements and points travel as canonical little-endian bytes, exactly the reference's to_repr() (utils.rs:10-14)
using Fr = std::array<uint8_t, 32>;   // scalar in [0, r)
using Fq = std::array<uint8_t, 32>;   // coordinate in [0, p)
struct G1Affine {
  Fq x{}, y{};
  bool infinity = false;              // the reference's points_to_bytes panics on it (lib.rs:58)
};

// Jacobian result (≙ C::Curve), z = 0 <=> identity
struct G1 {
  std::array<uint8_t, 96> xyz{};
  bool is_identity() const {
    for (int i = 64; i < 96; i++)
      if (xyz[i]) return false;
    return true;
  }
  G1Affine to_affine() const {  // ≙ Curve::to_affine
    G1Affine a;
    std::array<uint8_t, 64> xy{};
    const int r = msm_hip_g1_to_affine_bn254(xyz.data(), xy.data());
    if (r < 0) throw Error(r, "msm_hip_g1_to_affine_bn254");
    a.infinity = r == 1;
    std::memcpy(a.x.data(), xy.data(), 32);
    std::memcpy(a.y.data(), xy.data() + 32, 32);
    return a;
  }
  bool operator==(const G1& o) const {  // projective equality, as G1's PartialEq
    const G1Affine a = to_affine(), b = o.to_affine();
    return a.infinity == b.infinity && a.x == b.x && a.y == b.y;
  }
  bool operator!=(const G1& o) const { return !(*this == o); }
};

/// Convert scalars to bytes (src/lib.rs:50-52)
inline std::vector<uint8_t> scalars_to_bytes(const std::vector<Fr>& v) {
  std::vector<uint8_t> out(v.size() * 32);
  for (size_t i = 0; i < v.size(); i++) std::memcpy(out.data() + 32 * i, v[i].data(), 32);
  return out;
}

/// Convert points to bytes as [x0, y0, x1, y1, ...] (src/lib.rs:55-65); the point at infinity has no coordinates
inline std::vector<uint8_t> points_to_bytes(const std::vector<G1Affine>& g) {
  std::vector<uint8_t> out(g.size() * 64);
  for (size_t i = 0; i < g.size(); i++) {
    if (g[i].infinity) throw std::invalid_argument("points_to_bytes: point at infinity has no coordinates (src/lib.rs:58)");
    std::memcpy(out.data() + 64 * i, g[i].x.data(), 32);
    std::memcpy(out.data() + 64 * i + 32, g[i].y.data(), 32);
  }
  return out;
}

/// Persistent engine on one GPU (replaces the per-call device creation of src/cuzk/msm.rs:88-94)
class MsmContext {
 public:
  explicit MsmContext(int device = 0) { check(msm_hip_ctx_create(&ctx_, device), "msm_hip_ctx_create"); }
  ~MsmContext() { msm_hip_ctx_destroy(ctx_); }
  MsmContext(const MsmContext&) = delete;
  MsmContext& operator=(const MsmContext&) = delete;

  void set_bases(const std::vector<G1Affine>& g, bool check_on_curve = false, uint32_t more_flags = 0) {
    // more_flags: MSM_HIP_BASES_ENDOMORPHISM (half-length scalars, 2 x the base memory) or MSM_HIP_BASES_PRECOMPUTE (fixed-base tables)
    const std::vector<uint8_t> b = points_to_bytes(g);
    check(msm_hip_set_bases(ctx_, b.data(), g.size(), (check_on_curve ? MSM_HIP_CHECK_ON_CURVE : 0u) | more_flags), "msm_hip_set_bases");
  }
  /// Raw forms for callers whose field elements already sit in memory as bytes: `flags` as in msm_hip_set_bases
  /// (e.g. MSM_HIP_BASES_MONT256 for 4 x 64-bit Montgomery limbs); scalars_mont256(true) switches the scalar format likewise.
  void set_bases_bytes(const uint8_t* xy, size_t n, uint32_t flags = 0) {
    check(msm_hip_set_bases(ctx_, xy, n, flags), "msm_hip_set_bases");
  }
  void scalars_mont256(bool on) {
    check(msm_hip_set_scalar_format(ctx_, on ? MSM_HIP_SCALARS_MONT256 : MSM_HIP_SCALARS_CANONICAL), "msm_hip_set_scalar_format");
  }
  G1 msm_bytes(const uint8_t* scalars, size_t n) {
    G1 r;
    check(msm_hip_run(ctx_, scalars, n, r.xyz.data()), "msm_hip_run");
    return r;
  }
  G1 msm(const std::vector<Fr>& v) {
    const std::vector<uint8_t> b = scalars_to_bytes(v);
    G1 r;
    check(msm_hip_run(ctx_, b.data(), v.size(), r.xyz.data()), "msm_hip_run");
    return r;
  }
  /// Many MSMs over the resident bases (BASELINE config 5), pipelined inside the library: one result per scalar vector
  std::vector<G1> msm_batch(const std::vector<std::vector<Fr>>& vs) {
    if (vs.empty()) return {};
    const size_t n = vs[0].size();
    std::vector<uint8_t> all;
    all.reserve(vs.size() * n * 32);
    for (const auto& v : vs) {
      if (v.size() != n) throw std::invalid_argument("msm_batch: scalar vectors differ in length");
      const std::vector<uint8_t> b = scalars_to_bytes(v);
      all.insert(all.end(), b.begin(), b.end());
    }
    std::vector<uint8_t> out(96 * vs.size());
    check(msm_hip_run_batch(ctx_, all.data(), n, vs.size(), out.data()), "msm_hip_run_batch");
    std::vector<G1> r(vs.size());
    for (size_t k = 0; k < vs.size(); k++) std::memcpy(r[k].xyz.data(), out.data() + 96 * k, 96);
    return r;
  }
  msm_hip_ctx* raw() { return ctx_; }

 private:
  msm_hip_ctx* ctx_ = nullptr;
};

/// Several GPUs of one node from one process (msm_hip_mgpu_*): every MSM's windows are sharded over the devices, the window sums gathered
/// (RCCL, or pinned buffers when device ids repeat) and combined on the host.  `msm_stream` is the throughput form: groups of `group_size()`
/// scalar vectors per launch, up to three launches in flight, one result per vector -- what a prover with a queue of jobs over one set of
/// bases drives (the caller of src/lib.rs:76-82 many times over; the reference creates its device per call, src/cuzk/msm.rs:88-94).
class MultiGpuMsm {
 public:
  explicit MultiGpuMsm(const std::vector<int>& devices, uint32_t gather = MSM_HIP_MGPU_GATHER_AUTO) {
    check(msm_hip_mgpu_create(&m_, devices.data(), (int)devices.size(), gather), "msm_hip_mgpu_create");
  }
  ~MultiGpuMsm() { msm_hip_mgpu_destroy(m_); }
  MultiGpuMsm(const MultiGpuMsm&) = delete;
  MultiGpuMsm& operator=(const MultiGpuMsm&) = delete;

  void set_bases(const std::vector<G1Affine>& g, uint32_t flags = 0) {  // replicated on every device
    const std::vector<uint8_t> b = points_to_bytes(g);
    check(msm_hip_mgpu_set_bases(m_, b.data(), g.size(), flags), "msm_hip_mgpu_set_bases");
  }
  int group_size() const { return msm_hip_mgpu_group_size(m_); }
  bool uses_rccl() const { return msm_hip_mgpu_uses_rccl(m_) == 1; }
  G1 msm(const std::vector<Fr>& v) {
    const std::vector<uint8_t> b = scalars_to_bytes(v);
    G1 r;
    check(msm_hip_mgpu_run(m_, b.data(), v.size(), r.xyz.data()), "msm_hip_mgpu_run");
    return r;
  }
  std::vector<G1> msm_stream(const std::vector<std::vector<Fr>>& jobs) {
    if (jobs.empty()) return {};
    constexpr size_t IN_FLIGHT = 3;
    const size_t n = jobs[0].size(), g = (size_t)group_size();
    std::vector<std::vector<uint8_t>> groups;  // the library reads a group's bytes until its finish
    for (size_t first = 0; first < jobs.size(); first += g) {
      std::vector<uint8_t> blob;
      for (size_t k = first; k < jobs.size() && k < first + g; k++) {
        if (jobs[k].size() != n) throw std::invalid_argument("msm_stream: scalar vectors differ in length");
        const std::vector<uint8_t> b = scalars_to_bytes(jobs[k]);
        blob.insert(blob.end(), b.begin(), b.end());
      }
      groups.push_back(std::move(blob));
    }
    std::vector<G1> out;
    std::vector<uint8_t> xyz(96 * g);
    size_t launched = 0, finished = 0;  // groups launched / collected
    try {
      for (size_t i = 0; i < groups.size() + IN_FLIGHT; i++) {
        if (i >= IN_FLIGHT) {
          const size_t k = i - IN_FLIGHT, nvec = groups[k].size() / (32 * n);
          finished = k + 1;  // (finish leaves the slot free whatever it returns)
          check(msm_hip_mgpu_finish_batch(m_, (int)(k % MSM_HIP_NUM_SLOTS), xyz.data()), "msm_hip_mgpu_finish_batch");
          for (size_t v = 0; v < nvec; v++) {
            G1 r;
            std::memcpy(r.xyz.data(), xyz.data() + 96 * v, 96);
            out.push_back(r);
          }
        }
        if (i < groups.size()) {
          check(msm_hip_mgpu_launch_batch(m_, groups[i].data(), n, (int)(groups[i].size() / (32 * n)), (int)(i % MSM_HIP_NUM_SLOTS)),
                "msm_hip_mgpu_launch_batch");
          launched = i + 1;
        }
      }
    } catch (...) {
      // the launches still in flight read `groups` (pageable uploads on the devices' host threads): collect every one of them before the
      // buffers go out of scope
      for (size_t k = finished; k < launched; k++) (void)msm_hip_mgpu_finish_batch(m_, (int)(k % MSM_HIP_NUM_SLOTS), xyz.data());
      throw;
    }
    return out;
  }

 private:
  msm_hip_mgpu* m_ = nullptr;
};

/// ≙ compute_msm (src/cuzk/msm.rs:75): one-shot MSM including device set-up and base upload
inline G1 compute_msm(const std::vector<G1Affine>& points, const std::vector<Fr>& scalars) {
  if (points.size() != scalars.size()) throw std::invalid_argument("compute_msm: points and scalars differ in length");
  const std::vector<uint8_t> pb = points_to_bytes(points), sb = scalars_to_bytes(scalars);
  G1 r;
  check(msm_hip_msm_bn254_g1(pb.data(), sb.data(), scalars.size(), r.xyz.data()), "msm_hip_msm_bn254_g1");
  return r;
}

/// ≙ run_webgpu_msm (src/lib.rs:76-82); the name is the reference's, the device is an MI355X
inline G1 run_webgpu_msm(const std::vector<G1Affine>& g, const std::vector<Fr>& v) { return compute_msm(g, v); }

}  // namespace msm_webgpu
