// The per-device host thread of the multi-GPU object (msm_mgpu.h).  HIP-free on purpose: together with host_pool.h this is all of the
// product's own multi-threaded host code, and tests/host_harness/threads_harness.cpp compiles both under ThreadSanitizer and
// AddressSanitizer + UBSan on the CPU (tests/test_host_threads.py; no GPU sanitizers exist on this pool).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace {

// One persistent host thread per device: runs the closures posted to it in order.  H2D copies from pageable memory block their
// thread, and a launch is ~100 us of HIP calls: with a thread per device the devices' uploads and launches proceed side by side,
// and the caller's thread stays free (round 2 spawned the threads per call).
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
  uint64_t post(std::function<void()> f) {  // returns the ticket wait() takes
    std::lock_guard<std::mutex> lk(mu_);
    q_.push_back(std::move(f));
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
      cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
      if (q_.empty()) return;  // stop requested and nothing left to run
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
  std::thread th_;  // last: the thread starts when every other member exists
};

}  // namespace
