// Multi-GPU MSM behind the C ABI (include/msm_hip.h, msm_hip_mgpu_*): ONE host process drives one engine context per GPU.
//
// The reference is single-device (src/cuzk/msm.rs:88-94 creates one wgpu device per call); BASELINE.json's north star shards
// the independent Pippenger windows over the GPUs of a node "with a final RCCL gather/reduce of partial sums over xGMI".
//   one MSM      : device d computes the window sums of its contiguous window range (msm_hip_window_range; bases replicated,
//                  every device gets all scalars), the ranges' sums are gathered -- ncclAllGather of per x 96 B per device over
//                  RCCL (librccl is loaded at run time), or through each slot's pinned result buffer -- and the host window
//                  combine (src/cuzk/msm.rs:411-416) runs ONCE.
//   many MSMs    : whole MSMs are dealt out contiguously (BASELINE config 5); no exchange at all.
// Included by msm_hip.hip (same translation unit: it uses the context internals).
#pragma once
#include <dlfcn.h>

#include <thread>
#include <vector>

namespace {

constexpr int MGPU_MAX = 16;  // at most one device per window

// the handful of RCCL entry points the gather needs, resolved at run time so that libmsm_hip.so has no link-time dependency on
// librccl (a process that already holds one -- e.g. PyTorch's -- keeps using that one)
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather;
  }
};
constexpr int NCCL_UINT8 = 1;  // ncclUint8 (rccl.h)

}  // namespace

struct msm_hip_mgpu {
  int curve = MSM_HIP_CURVE_BN254_G1;
  int n = 0;
  int device[MGPU_MAX] = {};
  msm_hip_ctx* ctx[MGPU_MAX] = {};
  bool rccl = false;
  RcclApi api;
  void* comm[MGPU_MAX] = {};
  uint8_t* d_send[MGPU_MAX] = {};    // per x 96 B: this device's window sums, padded to the largest share
  uint8_t* d_gather[MGPU_MAX] = {};  // n x per x 96 B: every device's sums after the all-gather
  uint8_t* h_gather = nullptr;       // pinned copy of device 0's gather buffer
};

// run f(d) for every device on its own host thread (device 0 on the caller's): H2D copies from pageable memory block their
// thread, so the devices' uploads and launches proceed side by side
namespace {
template <typename F>
int mgpu_for_each(msm_hip_mgpu* m, F f) {
  std::vector<int> rc(m->n, MSM_HIP_OK);
  std::vector<std::thread> th;
  for (int d = 1; d < m->n; d++) th.emplace_back([&, d] { rc[d] = f(d); });
  rc[0] = f(0);
  for (std::thread& t : th) t.join();
  for (int d = 0; d < m->n; d++)
    if (rc[d]) return rc[d];
  return MSM_HIP_OK;
}

}  // namespace

extern "C" {

int msm_hip_window_range(int rank, int world, int num_windows, int* w_begin, int* w_end) {
  if (!w_begin || !w_end || world < 1 || rank < 0 || rank >= world || num_windows < 0) return MSM_HIP_ERR_INVALID_ARG;
  const int base = num_windows / world, extra = num_windows % world;  // the first `extra` ranks take one more
  *w_begin = rank * base + (rank < extra ? rank : extra);
  *w_end = *w_begin + base + (rank < extra ? 1 : 0);
  return MSM_HIP_OK;
}

void msm_hip_mgpu_destroy(msm_hip_mgpu* m) {
  if (!m) return;
  for (int d = 0; d < m->n; d++) {
    if (m->ctx[d]) {
      DeviceGuard guard(m->device[d]);
      if (m->comm[d]) (void)m->api.CommDestroy(m->comm[d]);
      if (m->d_send[d]) (void)hipFree(m->d_send[d]);
      if (m->d_gather[d]) (void)hipFree(m->d_gather[d]);
    }
    msm_hip_ctx_destroy(m->ctx[d]);
  }
  if (m->h_gather) (void)hipHostFree(m->h_gather);
  delete m;
}

int msm_hip_mgpu_create(msm_hip_mgpu** out, const int* device_ids, int n_devices, uint32_t flags) {
  return msm_hip_mgpu_create_curve(out, device_ids, n_devices, flags, MSM_HIP_CURVE_BN254_G1);
}

int msm_hip_mgpu_create_curve(msm_hip_mgpu** out, const int* device_ids, int n_devices, uint32_t flags, int curve) {
  if (!out) return MSM_HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > MGPU_MAX || flags > MSM_HIP_MGPU_GATHER_RCCL) return MSM_HIP_ERR_INVALID_ARG;
  if (curve < 0 || curve >= MSM_HIP_NUM_CURVES) return MSM_HIP_ERR_INVALID_ARG;
  msm_hip_mgpu* m = new (std::nothrow) msm_hip_mgpu();
  if (!m) return MSM_HIP_ERR_OUT_OF_MEMORY;
  m->curve = curve;
  bool distinct = true;
  for (int d = 0; d < n_devices; d++) {
    m->device[d] = device_ids[d];
    for (int e = 0; e < d; e++) distinct = distinct && device_ids[e] != device_ids[d];
    int rc = msm_hip_ctx_create_curve(&m->ctx[d], device_ids[d], curve);
    m->n = d + 1;
    if (rc) {
      m->n = d;  // ctx[d] was not created
      msm_hip_mgpu_destroy(m);
      return rc;
    }
  }
  // gather transport: RCCL when asked for, or by default when there is more than one (distinct) device and librccl loads
  const bool want_rccl = flags == MSM_HIP_MGPU_GATHER_RCCL || (flags == MSM_HIP_MGPU_GATHER_AUTO && n_devices > 1 && distinct);
  if (want_rccl) {
    bool ok = distinct && m->api.load() && m->api.CommInitAll(m->comm, n_devices, m->device) == 0;
    const int per = (NWIN + n_devices - 1) / n_devices;
    for (int d = 0; ok && d < n_devices; d++) {
      DeviceGuard guard(m->device[d]);
      ok = guard.ok && hipMalloc((void**)&m->d_send[d], (size_t)per * 96) == hipSuccess &&
           hipMalloc((void**)&m->d_gather[d], (size_t)n_devices * per * 96) == hipSuccess &&
           hipMemset(m->d_send[d], 0, (size_t)per * 96) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    ok = ok && hipHostMalloc((void**)&m->h_gather, (size_t)n_devices * per * 96, hipHostMallocDefault) == hipSuccess;
    if (!ok && flags == MSM_HIP_MGPU_GATHER_RCCL) {
      msm_hip_mgpu_destroy(m);
      return MSM_HIP_ERR_HIP;
    }
    m->rccl = ok;  // AUTO: fall back to the pinned-buffer gather
  }
  *out = m;
  return MSM_HIP_OK;
}

int msm_hip_mgpu_device_count(const msm_hip_mgpu* m) { return m ? m->n : MSM_HIP_ERR_INVALID_ARG; }
int msm_hip_mgpu_uses_rccl(const msm_hip_mgpu* m) { return m ? (m->rccl ? 1 : 0) : MSM_HIP_ERR_INVALID_ARG; }

int msm_hip_mgpu_set_bases_bn254(msm_hip_mgpu* m, const uint8_t* xy_host, size_t n, uint32_t flags) {
  if (!m || (!xy_host && n)) return MSM_HIP_ERR_INVALID_ARG;
  return mgpu_for_each(m, [&](int d) { return msm_hip_set_bases_bn254(m->ctx[d], xy_host, n, flags); });  // replicated
}

int msm_hip_mgpu_run_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]) {
  if (!m || !out_xyz || (!scalars_host && n)) return MSM_HIP_ERR_INVALID_ARG;
  if (n == 0) {
    memset(out_xyz, 0, 96);
    return MSM_HIP_OK;
  }
  const int G = m->n, per = (NWIN + G - 1) / G;
  // 1. every device: all scalars up, its own window range through the pipeline (slot 0)
  int rc = mgpu_for_each(m, [&](int d) {
    int b, e;
    (void)msm_hip_window_range(d, G, NWIN, &b, &e);
    return launch_host_windows(m->ctx[d], scalars_host, n, b, e, 0, m->rccl ? m->d_send[d] : nullptr);
  });
  if (rc) {
    for (int d = 0; d < G; d++) drain_slots(m->ctx[d]);
    return rc;
  }
  // 2. gather the window sums
  uint8_t all[NWIN * 288];
  bool parts = false;  // the pinned-buffer path receives every window sum as its three parts (Slot::parts): the host adds them up
  if (m->rccl) {
    // in stream order behind each device's bucket reduce: one all-gather over all devices, then device 0's copy to the host
    bool ok = m->api.GroupStart() == 0;
    for (int d = 0; ok && d < G; d++) {
      DeviceGuard guard(m->device[d]);
      ok = guard.ok && m->api.AllGather(m->d_send[d], m->d_gather[d], (size_t)per * 96, NCCL_UINT8, m->comm[d], m->ctx[d]->reduce_stream[0]) == 0;
    }
    ok = (m->api.GroupEnd() == 0) && ok;
    if (ok) {
      DeviceGuard guard(m->device[0]);
      ok = guard.ok && hipMemcpyAsync(m->h_gather, m->d_gather[0], (size_t)G * per * 96, hipMemcpyDeviceToHost, m->ctx[0]->reduce_stream[0]) == hipSuccess &&
           hipStreamSynchronize(m->ctx[0]->reduce_stream[0]) == hipSuccess;
    }
    for (int d = 0; d < G; d++) {  // error words; also leaves every slot collected
      const int r = msm_hip_slot_sync(m->ctx[d], 0);
      if (r && !rc) rc = r;
      if (d > 0) {
        DeviceGuard guard(m->device[d]);
        if (hipStreamSynchronize(m->ctx[d]->reduce_stream[0]) != hipSuccess) ok = false;  // its part of the collective
      }
    }
    if (rc) return rc;
    if (!ok) return MSM_HIP_ERR_HIP;
    for (int d = 0; d < G; d++) {
      int b, e;
      (void)msm_hip_window_range(d, G, NWIN, &b, &e);
      memcpy(all + (size_t)b * 96, m->h_gather + (size_t)d * per * 96, (size_t)(e - b) * 96);
    }
  } else {
    for (int d = 0; d < G; d++) {
      const int r = msm_hip_slot_sync(m->ctx[d], 0);
      if (r && !rc) rc = r;
      int b, e;
      (void)msm_hip_window_range(d, G, NWIN, &b, &e);
      const Slot& sl = m->ctx[d]->slot[0];
      if (d == 0) parts = sl.parts;
      if (sl.parts != parts) return MSM_HIP_ERR_HIP;  // (all contexts share the debug setting: cannot happen)
      const size_t rec = parts ? 288 : 96;
      memcpy(all + (size_t)b * rec, sl.h_wsums, (size_t)(e - b) * rec);
    }
    if (rc) return rc;
  }
  // 3. ONE host window combine
  if (parts) return curve_ops(m->curve)->combine_window_parts(all, NWIN, WBITS, out_xyz) ? MSM_HIP_OK : MSM_HIP_ERR_NONCANONICAL;
  return msm_hip_combine_windows_curve(m->curve, all, NWIN, out_xyz);
}

int msm_hip_mgpu_run_batch_bn254(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz) {
  if (!m || (!out_xyz && batch) || (!scalars_host && n && batch)) return MSM_HIP_ERR_INVALID_ARG;
  if (batch > (size_t)1 << 30) return MSM_HIP_ERR_INVALID_ARG;
  return mgpu_for_each(m, [&](int d) {
    int b, e;
    (void)msm_hip_window_range(d, m->n, (int)batch, &b, &e);
    if (e == b) return (int)MSM_HIP_OK;
    return msm_hip_run_batch_bn254(m->ctx[d], scalars_host + (size_t)b * n * 32, n, (size_t)(e - b), out_xyz + (size_t)b * 96);
  });
}

}  // extern "C"
