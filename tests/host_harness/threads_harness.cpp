// CPU harness for the product's multi-threaded host code -- csrc/host_worker.h (DeviceWorker: the per-device thread of msm_hip_mgpu_*) and
// csrc/host_pool.h (CombinePool: the host window combines of a launch, side by side) -- compiled by tests/test_host_threads.py under
// ThreadSanitizer and under AddressSanitizer + UBSan (no GPU sanitizers exist on this pool; these two headers and host_g1.h are HIP-free).
//
//   threads_harness <vectors.bin> [rounds]
// vectors.bin (written by the test from the ORACLE's Horner, oracle/bn254.c: oracle_horner): u32 cases, u32 windows, then per case
// windows x 96 B Jacobian window sums followed by the expected 96 B combined result (compared as affine points).
//
// What runs:
//   1. DeviceWorker: several poster threads post closures to several workers and wait on their tickets; every closure mutates plain
//      (non-atomic) per-worker state, so a missing lock or a broken happens-before in post / wait / loop is a data race TSan reports;
//      workers are destroyed with closures still queued (the destructor runs them, then joins).
//   2. CombinePool: concurrent callers of combine_pool().run (a busy pool makes the second caller run inline); every index exactly once.
//   3. the window combine of the mgpu / batch finish: combine_windows for every case from several threads through the pool, checked
//      against the oracle's result.
// Exit code 0 and "ok" on success.  -DHARNESS_BREAK_WORKER_LOCK removes the lock around the worker's queue push (the test checks that TSan
// then fails the run: the harness can see what it is meant to see).
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#ifdef HARNESS_BREAK_WORKER_LOCK
// the same class with the push left unlocked: a deliberately broken copy, compiled only to prove that the sanitizer run is not vacuous
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
namespace {
class DeviceWorker {
 public:
  DeviceWorker() : th_([this] { loop(); }) {}
  ~DeviceWorker() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  uint64_t post(std::function<void()> f) {
    q_.push_back(std::move(f));  // <-- no lock_guard
    cv_.notify_all();
    return ++posted_;
  }
  void wait(uint64_t ticket) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return done_ >= ticket; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait_for(lk, std::chrono::milliseconds(1), [&] { return stop_ || !q_.empty(); });
      if (q_.empty()) {
        if (stop_) return;
        continue;
      }
      std::function<void()> f = std::move(q_.front());
      q_.pop_front();
      lk.unlock();
      f();
      lk.lock();
      done_++;
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
  uint64_t posted_ = 0, done_ = 0;
  bool stop_ = false;
  std::thread th_;
};
}  // namespace
#else
#include "host_worker.h"
#endif
#include "host_pool.h"
#include "host_g1.h"

namespace {

int fail(const char* what) {
  fprintf(stderr, "threads_harness: %s\n", what);
  return 1;
}

// 1. DeviceWorker under several posters
int hammer_workers(int rounds) {
  constexpr int WORKERS = 4, POSTERS = 3, PER_POSTER = 200;
  for (int r = 0; r < rounds; r++) {
    std::vector<std::unique_ptr<DeviceWorker>> w;
    for (int i = 0; i < WORKERS; i++) w.emplace_back(new DeviceWorker());
    // plain state, touched only from the worker's own thread (and read by a poster after wait(): post -> run -> wait must order them)
    struct State {
      long counter = 0;
      std::vector<int> log;
    } state[WORKERS];
    std::atomic<long> posted_total{0};
    std::vector<std::thread> posters;
    for (int p = 0; p < POSTERS; p++)
      posters.emplace_back([&, p] {
        for (int k = 0; k < PER_POSTER; k++) {
          const int d = (p + k) % WORKERS;
          const bool waited = k % 7 == 0;
          long seen_before = -1;  // written by the closure only when this poster waits for it (else the closure must not touch this stack frame)
          long* seen = waited ? &seen_before : nullptr;
          const uint64_t t = w[d]->post([&state, d, p, k, seen] {
            if (seen) *seen = state[d].counter;
            state[d].counter++;
            state[d].log.push_back(p * 1000 + k);
          });
          posted_total++;
          if (waited) {
            w[d]->wait(t);
            if (seen_before < 0) abort();  // the closure has run and its write is visible here (post -> run -> wait orders them: TSan checks)
          }
        }
      });
    for (std::thread& t : posters) t.join();
    // destroy with work possibly still queued: the destructor drains the queue, then joins
    for (int i = 0; i < WORKERS; i++) w[i]->post([&state, i] { state[i].counter += 1000000; });
    w.clear();
    long total = 0;
    for (int i = 0; i < WORKERS; i++) {
      if (state[i].counter < 1000000) return fail("a closure queued at destruction did not run");
      total += state[i].counter - 1000000;
      if ((long)state[i].log.size() != state[i].counter - 1000000) return fail("worker log and counter disagree");
    }
    if (total != posted_total.load()) return fail("closures lost or run twice");
  }
  return 0;
}

// 2. CombinePool under concurrent callers
int hammer_pool(int rounds) {
  constexpr int CALLERS = 4;
  std::atomic<int> bad{0};
  std::vector<std::thread> callers;
  for (int c = 0; c < CALLERS; c++)
    callers.emplace_back([&, c] {
      for (int r = 0; r < rounds * 50; r++) {
        const int count = 1 + (c * 7 + r) % 23;
        std::vector<int> hits(count, 0);  // plain ints: every index is written by exactly one thread of the job
        long sum = 0;
        std::mutex m;
        combine_pool().run(count, [&](int i) {
          hits[i]++;
          std::lock_guard<std::mutex> lk(m);
          sum += i;
        });
        for (int i = 0; i < count; i++)
          if (hits[i] != 1) bad++;
        if (sum != (long)count * (count - 1) / 2) bad++;
      }
    });
  for (std::thread& t : callers) t.join();
  return bad.load() ? fail("CombinePool ran an index not exactly once") : 0;
}

// 3. window combines against the oracle's Horner
int combine_cases(const char* path, int rounds) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail("cannot open the vector file");
  uint32_t cases = 0, windows = 0;
  if (fread(&cases, 4, 1, f) != 1 || fread(&windows, 4, 1, f) != 1 || cases == 0 || windows == 0 || windows > 16) return fail("bad vector header");
  const size_t rec = (size_t)windows * 96 + 96;
  std::vector<uint8_t> data((size_t)cases * rec);
  if (fread(data.data(), 1, data.size(), f) != data.size()) return fail("short vector file");
  fclose(f);
  std::atomic<int> bad{0};
  auto check_all = [&](int stride_off) {
    std::vector<uint8_t> out((size_t)cases * 96);
    combine_pool().run((int)cases, [&](int v) {
      if (!bn254::host::combine_windows(data.data() + (size_t)v * rec, (int)windows, 16, out.data() + (size_t)v * 96)) bad++;
    });
    for (uint32_t v = 0; v < cases; v++) {
      uint8_t got[64], want[64];
      const int a = bn254::host::to_affine64(out.data() + (size_t)v * 96, got);
      const int b = bn254::host::to_affine64(data.data() + (size_t)v * rec + (size_t)windows * 96, want);
      if (a < 0 || a != b || memcmp(got, want, 64) != 0) bad++;
    }
    (void)stride_off;
  };
  std::vector<std::thread> callers;
  for (int c = 0; c < 3; c++)
    callers.emplace_back([&, c] {
      for (int r = 0; r < rounds; r++) check_all(c);
    });
  for (std::thread& t : callers) t.join();
  return bad.load() ? fail("a window combine differs from the oracle's Horner") : 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return fail("usage: threads_harness <vectors.bin> [rounds]");
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  if (int rc = hammer_workers(rounds)) return rc;
  if (int rc = hammer_pool(rounds)) return rc;
  if (int rc = combine_cases(argv[1], rounds)) return rc;
  puts("ok");
  return 0;
}
