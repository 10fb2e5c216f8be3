// Multi-GPU MSM behind the C ABI (include/msm_hip.h, msm_hip_mgpu_*): ONE host process drives one engine context per GPU.
//
// The reference is single-device (src/cuzk/msm.rs:88-94 creates one wgpu device per call); BASELINE.json's north star shards
// the independent Pippenger windows over the GPUs of a node "with a final RCCL gather/reduce of partial sums over xGMI".
//
//   window-sharded launches (msm_hip_mgpu_launch_batch_* / msm_hip_mgpu_finish_batch; msm_hip_mgpu_run = one vector, slot 0):
//     device d computes the window sums of ITS window range for every scalar vector of the launch -- `nvec` MSMs' shares go through
//     one kernel sequence per device, because one MSM's share (2 of 16 windows at 8 GPUs) cannot fill a GPU -- into result slot
//     `slot` of its context; the shares are gathered with ONE ncclAllGather per launch, in stream order behind each device's bucket
//     reduce (librccl is loaded at run time), or through the slots' pinned buffers; one host window combine per MSM
//     (src/cuzk/msm.rs:411-416), spread over the host pool.  With endomorphism bases the windows are the 8 half-length ones.
//     Launches are asynchronous: every device has a persistent host thread that issues its HIP / RCCL calls, so the caller can keep
//     several result slots in flight (launch k + 1 and k + 2 run while launch k is gathered and combined).
//   many whole MSMs (msm_hip_mgpu_run_batch): dealt out contiguously (BASELINE config 5); no exchange at all.
// Included by msm_hip.hip (same translation unit: it uses the context internals).
#pragma once
#include <dlfcn.h>

#include <vector>

#include "host_worker.h"

namespace {

constexpr int MGPU_MAX = 16;  // at most one device per window

// the handful of RCCL entry points the gather needs, resolved at run time so that libmsm_hip.so has no link-time dependency on
// librccl (a process that already holds one -- e.g. PyTorch's -- keeps using that one)
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*CommAbort)(void* comm) = nullptr;  // (optional: only used to tear down a communicator whose collective could not be issued)
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
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather;
  }
};
constexpr int NCCL_UINT8 = 1;  // ncclUint8 (rccl.h)

// what a result slot of the multi-GPU object holds between launch and finish
struct MgpuSlot {
  bool pending = false;
  int nvec = 0;
  size_t n = 0;
  bool halves = false;           // the windows are the 8 half-length ones (endomorphism bases)
  int wide_bits = 0;             // != 0: the shares are the virtual windows of the wide fixed-base tables (records come in pairs: window sum, plain total)
  uint64_t ticket[MGPU_MAX] = {};  // the launch job of every device
  int rc[MGPU_MAX] = {};           // ... and its status
  bool launched[MGPU_MAX] = {};    // the device's context slot holds an unfinished launch
  uint8_t* d_send[MGPU_MAX] = {};    // 2 MAXLW x 96 B: this device's window sums [nvec][its windows] (wide shares: record pairs), then padding
  uint8_t* d_gather[MGPU_MAX] = {};  // n x 2 MAXLW x 96 B: every device's block after the all-gather
  uint8_t* h_gather = nullptr;       // pinned copy of device 0's gather buffer
  hipEvent_t gathered[MGPU_MAX] = {};  // device d's call of this launch's all-gather has completed (device 0: and h_gather is complete)
};

}  // namespace

struct msm_hip_mgpu {
  int curve = MSM_HIP_CURVE_BN254_G1;
  int n = 0;
  int device[MGPU_MAX] = {};
  msm_hip_ctx* ctx[MGPU_MAX] = {};
  DeviceWorker* worker[MGPU_MAX] = {};
  bool rccl = false;
  std::atomic<bool> broken{false};  // a device could not issue its call of a collective: the communicator is out of step, every later call fails
  std::atomic<int> fault_device{-1};  // test hook (msm_hip_mgpu_inject_fault): the next `fault_left` launches fail on this device
  std::atomic<int> fault_left{0};
  bool endo = false;  // the resident bases carry their endomorphism images: window-sharded launches use the 8 half-length windows
  int wide_bits = 0;  // the resident bases are wide fixed-base tables of this digit width: window-sharded launches share their 2^(C-16) virtual windows
  int wide_bits_choice = 0;  // msm_hip_mgpu_set_wide_bits (0: 19 -- as many virtual windows as a node has GPUs)
  RcclApi api;
  void* comm[MGPU_MAX] = {};
  hipStream_t gather_stream[MGPU_MAX] = {};  // the collective and the copy to the host: behind the slot's bucket reduce, beside the next one
  MgpuSlot slot[NSLOT];
};

namespace {
// run f(d) for every device on its worker thread and wait for all of them
template <typename F>
int mgpu_for_each(msm_hip_mgpu* m, F f) {
  int rc[MGPU_MAX] = {};
  uint64_t ticket[MGPU_MAX];
  for (int d = 0; d < m->n; d++) ticket[d] = m->worker[d]->post([&rc, &f, d] { rc[d] = f(d); });
  for (int d = 0; d < m->n; d++) m->worker[d]->wait(ticket[d]);
  for (int d = 0; d < m->n; d++)
    if (rc[d]) return rc[d];
  return MSM_HIP_OK;
}

// the windows the devices share: 16, the 8 half-length ones (endomorphism bases), or the virtual windows of wide tables
inline int mgpu_windows(bool halves, int wide_bits) { return wide_bits ? wide_vwin_of(wide_bits) : halves ? nwin_of(WBITS, true) : NWIN; }
inline int mgpu_per(const msm_hip_mgpu* m, bool halves, int wide_bits) { return (mgpu_windows(halves, wide_bits) + m->n - 1) / m->n; }

// device d's own work of a window-sharded launch: scalars up (host variant), its window range of every vector into context slot `k`
// (`launched`: the context slot now holds an unfinished launch)
int mgpu_enqueue_on_device(msm_hip_mgpu* m, int d, int k, const void* scalars, bool host_scalars, size_t n, int nvec, int b, int e, bool& launched) {
  MgpuSlot& ms = m->slot[k];
  msm_hip_ctx* ctx = m->ctx[d];
  const bool halves = ms.halves;
  int rc = MSM_HIP_OK;
  if (m->fault_device.load() == d && m->fault_left.load() > 0) {  // injected fault (tests): this device's launch fails before anything is queued
    m->fault_left--;
    return MSM_HIP_ERR_HIP;
  }
  const void* dev = scalars;
  if (host_scalars) {  // all vectors of the launch into the slot's staging buffer, on the copy stream
    ON_DEVICE(ctx);
    Slot& s = ctx->slot[k];
    if (s.pending) return MSM_HIP_ERR_SLOT_BUSY;
    if ((rc = setup_slot(ctx, s))) return rc;
    if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    const size_t count = (size_t)nvec * n;
    if (count > s.cap_host_scalars) {
      s.cap_host_scalars = 0;
      if ((rc = dev_alloc(ctx, s.d_host_scalars, count * 8))) return rc;
      s.cap_host_scalars = count;
    }
    HIP_TRY(ctx, hipMemcpyAsync(s.d_host_scalars, scalars, count * 32, hipMemcpyHostToDevice, ctx->copy_stream));
    HIP_TRY(ctx, hipEventRecord(s.staged, ctx->copy_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.staged, 0));
    dev = s.d_host_scalars;
  }
  void* sums = m->rccl ? ms.d_send[d] : nullptr;  // host gather: the sums leave through the context slot's pinned buffer
  if (ms.wide_bits) rc = launch_impl(ctx, dev, n, nvec, 0, wide_tables_of(ms.wide_bits), ms.wide_bits, k, sums, MODE_WIDE, b, e - b);
  else rc = launch_impl(ctx, dev, n, nvec, b, e, WBITS, k, sums, halves ? MODE_HALVES : MODE_PLAIN);
  launched = rc == MSM_HIP_OK;
  return rc;
}

// device d's part of a window-sharded launch (on its worker thread): its own work (above), then -- RCCL -- its call of the launch's
// all-gather, in stream order behind the slot.  The collective is issued WHATEVER happened before it: the devices' calls of one
// all-gather must pair up, and a device that returned early (slot busy, out of memory, a HIP error) would leave its peers' calls
// waiting for ever -- finish would hang -- and, with several slots in flight, pair its NEXT launch's call with their stale one (a
// slot would then receive another launch's window sums).  A failed device sends a zeroed block; finish reports its status.
int mgpu_launch_on_device(msm_hip_mgpu* m, int d, int k, const void* scalars, bool host_scalars, size_t n, int nvec) {
  MgpuSlot& ms = m->slot[k];
  msm_hip_ctx* ctx = m->ctx[d];
  const bool halves = ms.halves;
  const int W = mgpu_windows(halves, ms.wide_bits), rows = nvec * mgpu_per(m, halves, ms.wide_bits) * (ms.wide_bits ? 2 : 1);
  int b, e;
  (void)msm_hip_window_range(d, m->n, W, &b, &e);
  int rc = MSM_HIP_OK;
  bool launched = false;
  if (e > b) rc = mgpu_enqueue_on_device(m, d, k, scalars, host_scalars, n, nvec, b, e, launched);
  ms.launched[d] = launched;
  if (m->rccl) {
    auto collective = [&]() -> int {
      ON_DEVICE(ctx);
      hipStream_t gs = m->gather_stream[d];
      const size_t bytes = (size_t)rows * ctx->jb;
      if (launched) HIP_TRY(ctx, hipStreamWaitEvent(gs, ctx->slot[k].done, 0));
      else if (e > b) {
        // a failed launch sends zeros.  It may have failed AFTER queueing kernels that write the send block (on the context's main / reduce
        // streams): the fill is ordered behind everything those streams hold, or a late kernel would overwrite it while the collective reads
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (hipStream_t r : ctx->reduce_stream) HIP_TRY(ctx, hipStreamSynchronize(r));
        HIP_TRY(ctx, hipMemsetAsync(ms.d_send[d], 0, bytes, gs));  // (a device without windows sends the zeros of its creation)
      }
      if (m->api.AllGather(ms.d_send[d], ms.d_gather[d], bytes, NCCL_UINT8, m->comm[d], gs) != 0) return MSM_HIP_ERR_HIP;
      if (d == 0) HIP_TRY(ctx, hipMemcpyAsync(ms.h_gather, ms.d_gather[0], (size_t)m->n * bytes, hipMemcpyDeviceToHost, gs));
      HIP_TRY(ctx, hipEventRecord(ms.gathered[d], gs));
      return MSM_HIP_OK;
    };
    const int crc = collective();
    if (crc) {  // this device's call could not be issued: the communicator is out of step for good
      m->broken = true;
      if (!rc) rc = crc;
    }
  }
  return rc;
}

int mgpu_launch(msm_hip_mgpu* m, const void* const* per_device, const void* host, bool is_host, size_t n, int nvec, int k) {
  if (!m || k < 0 || k >= NSLOT || nvec < 1 || n > MAX_POINTS) return MSM_HIP_ERR_INVALID_ARG;
  if (m->broken) return MSM_HIP_ERR_HIP;  // (a collective could not be issued on some device earlier: destroy the object)
  MgpuSlot& ms = m->slot[k];
  if (ms.pending) return MSM_HIP_ERR_SLOT_BUSY;
  const bool halves = m->endo;
  if (nvec * mgpu_per(m, halves, m->wide_bits) > MAXLW) return MSM_HIP_ERR_INVALID_ARG;
  if (n && m->ctx[0]->n_bases == 0) return MSM_HIP_ERR_NO_BASES;
  if (n > m->ctx[0]->n_bases) return MSM_HIP_ERR_INVALID_ARG;
  ms.pending = true;
  ms.nvec = nvec;
  ms.n = n;
  ms.halves = halves;
  ms.wide_bits = m->wide_bits;
  for (int d = 0; d < m->n; d++) {
    ms.launched[d] = false;
    ms.rc[d] = MSM_HIP_OK;
    const void* sc = is_host ? host : per_device[d];
    ms.ticket[d] = m->worker[d]->post([m, d, k, sc, is_host, n, nvec] {
      m->slot[k].rc[d] = n ? mgpu_launch_on_device(m, d, k, sc, is_host, n, nvec) : MSM_HIP_OK;
    });
  }
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
  for (int k = 0; k < NSLOT; k++)  // unfinished launches: their jobs must have run before anything is torn down
    if (m->slot[k].pending)
      for (int d = 0; d < m->n; d++)
        if (m->worker[d]) m->worker[d]->wait(m->slot[k].ticket[d]);
  for (int d = 0; d < m->n; d++) {
    delete m->worker[d];  // (runs what is still queued, then joins)
    m->worker[d] = nullptr;
  }
  if (m->broken && m->api.CommAbort) {  // collectives that can never pair up: abort them instead of waiting
    for (int d = 0; d < m->n; d++)
      if (m->comm[d]) {
        DeviceGuard guard(m->device[d]);
        (void)m->api.CommAbort(m->comm[d]);
        m->comm[d] = nullptr;
      }
  }
  for (int d = 0; d < m->n; d++)  // every device's part of the collectives still in flight, before any communicator goes
    if (m->ctx[d] && m->gather_stream[d]) {
      DeviceGuard guard(m->device[d]);
      (void)hipStreamSynchronize(m->gather_stream[d]);
    }
  for (int d = 0; d < m->n; d++) {
    if (m->ctx[d]) {
      DeviceGuard guard(m->device[d]);
      if (m->comm[d]) (void)m->api.CommDestroy(m->comm[d]);
      for (MgpuSlot& ms : m->slot) {
        if (ms.d_send[d]) (void)hipFree(ms.d_send[d]);
        if (ms.d_gather[d]) (void)hipFree(ms.d_gather[d]);
        if (ms.gathered[d]) (void)hipEventDestroy(ms.gathered[d]);
      }
      if (m->gather_stream[d]) (void)hipStreamDestroy(m->gather_stream[d]);
    }
    msm_hip_ctx_destroy(m->ctx[d]);
  }
  for (MgpuSlot& ms : m->slot)
    if (ms.h_gather) (void)hipHostFree(ms.h_gather);
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
    m->worker[d] = new (std::nothrow) DeviceWorker();
    if (!m->worker[d]) {
      msm_hip_mgpu_destroy(m);
      return MSM_HIP_ERR_OUT_OF_MEMORY;
    }
  }
  // gather transport: RCCL when asked for, or by default when there is more than one (distinct) device and librccl loads
  const bool want_rccl = flags == MSM_HIP_MGPU_GATHER_RCCL || (flags == MSM_HIP_MGPU_GATHER_AUTO && n_devices > 1 && distinct);
  if (want_rccl) {
    bool ok = distinct && m->api.load() && m->api.CommInitAll(m->comm, n_devices, m->device) == 0;
    for (int d = 0; ok && d < n_devices; d++) {
      DeviceGuard guard(m->device[d]);
      ok = guard.ok && hipStreamCreateWithFlags(&m->gather_stream[d], hipStreamNonBlocking) == hipSuccess;
      for (int k = 0; ok && k < NSLOT; k++) {
        MgpuSlot& ms = m->slot[k];
        ok = hipMalloc((void**)&ms.d_send[d], (size_t)2 * MAXLW * MAX_JB) == hipSuccess &&
             hipMalloc((void**)&ms.d_gather[d], (size_t)n_devices * 2 * MAXLW * MAX_JB) == hipSuccess &&
             hipMemset(ms.d_send[d], 0, (size_t)2 * MAXLW * MAX_JB) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&ms.gathered[d], hipEventDisableTiming) == hipSuccess;
      }
      ok = ok && hipDeviceSynchronize() == hipSuccess;
    }
    for (int k = 0; ok && k < NSLOT; k++)
      ok = hipHostMalloc((void**)&m->slot[k].h_gather, (size_t)n_devices * 2 * MAXLW * MAX_JB, hipHostMallocDefault) == hipSuccess;
    if (!ok && flags == MSM_HIP_MGPU_GATHER_RCCL) {
      msm_hip_mgpu_destroy(m);
      return MSM_HIP_ERR_HIP;
    }
    m->rccl = ok;  // AUTO: fall back to the pinned-buffer gather
  }
  *out = m;
  return MSM_HIP_OK;
}

int msm_hip_mgpu_inject_fault(msm_hip_mgpu* m, int device_index, int launches) {
  if (!m || device_index < 0 || device_index >= m->n || launches < 0) return MSM_HIP_ERR_INVALID_ARG;
  m->fault_device = device_index;
  m->fault_left = launches;
  return MSM_HIP_OK;
}

int msm_hip_mgpu_device_count(const msm_hip_mgpu* m) { return m ? m->n : MSM_HIP_ERR_INVALID_ARG; }
int msm_hip_mgpu_uses_rccl(const msm_hip_mgpu* m) { return m ? (m->rccl ? 1 : 0) : MSM_HIP_ERR_INVALID_ARG; }

int msm_hip_mgpu_set_bases(msm_hip_mgpu* m, const uint8_t* xy_host, size_t n, uint32_t flags) {
  if (!m || (!xy_host && n)) return MSM_HIP_ERR_INVALID_ARG;
  for (const MgpuSlot& ms : m->slot)
    if (ms.pending) return MSM_HIP_ERR_SLOT_BUSY;
  // The mode the window-sharded launches run in follows the flags: the 8 half-length windows with the endomorphism images, the virtual windows
  // with wide tables, else the reference's 16 windows over the n plain records.  flags = 0 (the drop-in call shape) is resolved ONCE here to the
  // plain set -- a context's own default (the endomorphism images on the prime-order curves, msm_hip.hip: resolve_base_flags) would make every
  // device store 2n records and run k_endo_points for launches that then read only the first n.  Whole-MSM batches
  // (msm_hip_mgpu_run_batch) run in the mode the flags name.
  if (!(flags & (MSM_HIP_BASES_PRECOMPUTE | MSM_HIP_BASES_PRECOMPUTE_WIDE | MSM_HIP_BASES_ENDOMORPHISM | MSM_HIP_BASES_PLAIN))) flags |= MSM_HIP_BASES_PLAIN;
  if (flags & MSM_HIP_BASES_PRECOMPUTE_WIDE)  // shares of virtual windows: 19-bit digits unless asked otherwise (a context alone picks 17 / 20 by n: 2 or 16 virtual windows)
    for (int d = 0; d < m->n; d++) m->ctx[d]->wide_bits_choice = m->wide_bits_choice ? m->wide_bits_choice : 19;
  const int rc = mgpu_for_each(m, [&](int d) { return msm_hip_set_bases(m->ctx[d], xy_host, n, flags); });  // replicated
  m->endo = !rc && n && m->ctx[0]->endo;
  m->wide_bits = !rc && n ? m->ctx[0]->wide_bits : 0;
  return rc;
}

int msm_hip_mgpu_set_wide_bits(msm_hip_mgpu* m, int bits) {
  if (!m || (bits != 0 && (bits < 16 || bits > 20))) return MSM_HIP_ERR_INVALID_ARG;
  m->wide_bits_choice = bits;
  return MSM_HIP_OK;
}

int msm_hip_mgpu_group_size(const msm_hip_mgpu* m) {
  if (!m) return MSM_HIP_ERR_INVALID_ARG;
  const int per = mgpu_per(m, m->endo, m->wide_bits);
  int g = mgpu_windows(m->endo, m->wide_bits) / per;  // as many MSMs' shares as make up one MSM's worth of bucket sets per device
  return g < 1 ? 1 : g;
}

int msm_hip_mgpu_launch_batch(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, int nvec, int slot) {
  if (!scalars_host && n) return MSM_HIP_ERR_INVALID_ARG;
  return mgpu_launch(m, nullptr, scalars_host, true, n, nvec, slot);
}

int msm_hip_mgpu_launch_batch_device(msm_hip_mgpu* m, const void* const* scalars_dev, size_t n, int nvec, int slot) {
  if (!m || !scalars_dev) return MSM_HIP_ERR_INVALID_ARG;
  for (int d = 0; d < m->n; d++)
    if (!scalars_dev[d] && n) return MSM_HIP_ERR_INVALID_ARG;
  return mgpu_launch(m, scalars_dev, nullptr, false, n, nvec, slot);
}

int msm_hip_mgpu_finish_batch(msm_hip_mgpu* m, int slot, uint8_t* out_xyz) {
  if (!m || !out_xyz || slot < 0 || slot >= NSLOT) return MSM_HIP_ERR_INVALID_ARG;
  MgpuSlot& ms = m->slot[slot];
  if (!ms.pending) return MSM_HIP_ERR_INVALID_ARG;
  const int wide_bits = ms.wide_bits;
  const int G = m->n, nvec = ms.nvec, W = mgpu_windows(ms.halves, wide_bits), rows = nvec * mgpu_per(m, ms.halves, wide_bits) * (wide_bits ? 2 : 1);
  int rc = MSM_HIP_OK;
  // 1. every device's launch job has run; the first failure is the launch's status
  for (int d = 0; d < G; d++) {
    m->worker[d]->wait(ms.ticket[d]);
    if (ms.rc[d] && !rc) rc = ms.rc[d];
  }
  // 2. RCCL: every device's call of THIS launch's all-gather has completed (device 0: and its copy of the gathered blocks) -- the slot's
  //    send / gather buffers are free again.  The calls were issued in lock-step whatever the devices' own launches did, so this wait
  //    ends even when the launch failed somewhere; only a collective that could not be ISSUED (broken) may never complete: not waited for.
  //    (Per-slot events: the gather streams also carry the all-gathers of the launches behind this one.)
  bool hip_ok = true;
  if (m->rccl && ms.n && !m->broken)
    for (int d = 0; d < G; d++) {
      DeviceGuard guard(m->device[d]);
      if (!guard.ok || hipEventSynchronize(ms.gathered[d]) != hipSuccess) hip_ok = false;
    }
  // 3. collect every context slot (error words; leaves the slots free whatever happened) -- on the device's own worker thread, which
  //    is the only thread that touches its context while launches are in flight
  bool parts = false;
  {
    int r[MGPU_MAX] = {};
    bool p[MGPU_MAX] = {};
    uint64_t ticket[MGPU_MAX] = {};
    for (int d = 0; d < G; d++)
      if (ms.launched[d]) ticket[d] = m->worker[d]->post([m, d, slot, &r, &p] {
        r[d] = msm_hip_slot_sync(m->ctx[d], slot);
        p[d] = m->ctx[d]->slot[slot].parts;
      });
    for (int d = 0; d < G; d++) {
      if (!ms.launched[d]) continue;
      m->worker[d]->wait(ticket[d]);
      if (r[d] && !rc) rc = r[d];
      parts = parts || p[d];
    }
  }
  if (m->broken && !rc) rc = MSM_HIP_ERR_HIP;
  ms.pending = false;
  if (rc) return rc;
  if (!hip_ok) return MSM_HIP_ERR_HIP;
  if (ms.n == 0) {  // nothing was launched anywhere
    memset(out_xyz, 0, (size_t)nvec * m->ctx[0]->jb);
    return MSM_HIP_OK;
  }
  // 4. one host window combine per MSM, side by side on the host pool
  const size_t jb = m->ctx[0]->jb;
  // (one vector per launch through the pinned buffers: the windows arrive as bit-plane sums; shares of virtual windows: as record pairs)
  const size_t rec = wide_bits ? 2 * jb : parts ? PLANES_PER_WINDOW * jb : jb;
  const CurveOps* ops = curve_ops(m->curve);
  std::atomic<bool> ok{true};
  combine_pool().run(nvec, [&](int v) {
    uint8_t all[NWIN * PLANES_PER_WINDOW * MAX_JB], sums[NWIN * MAX_JB];
    for (int d = 0; d < G; d++) {
      int b, e;
      (void)msm_hip_window_range(d, G, W, &b, &e);
      if (e == b) continue;
      const uint8_t* block = m->rccl ? ms.h_gather + (size_t)d * rows * jb : m->ctx[d]->slot[slot].h_wsums;
      memcpy(all + (size_t)b * rec, block + (size_t)v * (e - b) * rec, (size_t)(e - b) * rec);
    }
    const uint8_t* records = all;
    if (wide_bits) {  // sum_hi W_hi + 2^15 sum_hi hi TC_hi over the gathered (W_hi, TC_hi) pairs
      if (!ops->combine_wide_pairs(all, W, out_xyz + (size_t)v * jb)) ok = false;
      return;
    }
    if (parts) {
      for (int w = 0; w < W; w++)
        if (!ops->window_from_planes(all + (size_t)w * rec, sums + jb * (size_t)w)) ok = false;
      records = sums;
    }
    if (!ops->combine_windows(records, W, WBITS, out_xyz + (size_t)v * jb)) ok = false;
  });
  return ok ? MSM_HIP_OK : MSM_HIP_ERR_NONCANONICAL;
}

int msm_hip_mgpu_run(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, uint8_t out_xyz[96]) {
  if (!m || !out_xyz || (!scalars_host && n)) return MSM_HIP_ERR_INVALID_ARG;
  if (n == 0) {
    memset(out_xyz, 0, m->ctx[0]->jb);
    return MSM_HIP_OK;
  }
  int rc = msm_hip_mgpu_launch_batch(m, scalars_host, n, 1, 0);
  if (rc) return rc;
  return msm_hip_mgpu_finish_batch(m, 0, out_xyz);
}

int msm_hip_mgpu_run_batch(msm_hip_mgpu* m, const uint8_t* scalars_host, size_t n, size_t batch, uint8_t* out_xyz) {
  if (!m || (!out_xyz && batch) || (!scalars_host && n && batch)) return MSM_HIP_ERR_INVALID_ARG;
  if (batch > (size_t)1 << 30) return MSM_HIP_ERR_INVALID_ARG;
  for (const MgpuSlot& ms : m->slot)
    if (ms.pending) return MSM_HIP_ERR_SLOT_BUSY;
  return mgpu_for_each(m, [&](int d) {
    int b, e;
    (void)msm_hip_window_range(d, m->n, (int)batch, &b, &e);
    if (e == b) return (int)MSM_HIP_OK;
    return msm_hip_run_batch(m->ctx[d], scalars_host + (size_t)b * n * 32, n, (size_t)(e - b), out_xyz + (size_t)b * m->ctx[d]->jb);
  });
}

}  // extern "C"
