// A small persistent pool of host threads for the independent host-side window combines of one launch (src/cuzk/msm.rs:411-416 per MSM:
// 240 dependent doublings, ~47 us on one core).  A launch that carries several MSMs (grouped small MSMs, the window shares of 8 MSMs
// in the multi-GPU pipeline) has that many independent chains; run serially they sit on the caller's critical path (8 x 47 us behind
// every launch, all of it exposed behind the last one).  The pool runs f(0) .. f(count - 1) on up to 8 threads including the caller
// and returns when all are done.  Created on first use; one job at a time (a second caller that finds it busy runs its job inline).
#pragma once
#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

class CombinePool {
 public:
  template <typename F>
  void run(int count, F f) {
    if (count <= 0) return;
    std::unique_lock<std::mutex> job_lock(job_mu_, std::try_to_lock);
    if (count == 1 || !job_lock.owns_lock() || !start()) {
      for (int i = 0; i < count; i++) f(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = [&f](int i) { f(i); };
      total_ = count;
      next_ = 0;
      finished_ = 0;
      generation_++;
    }
    cv_work_.notify_all();
    work();  // the caller takes items too
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return finished_ == total_; });
    total_ = 0;
    fn_ = nullptr;
  }
  ~CombinePool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (std::thread& t : threads_) t.join();
  }

 private:
  bool start() {  // under job_mu_
    if (!threads_.empty()) return true;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = hw > 1 ? (int)(hw - 1 < 7 ? hw - 1 : 7) : 0;
    if (const char* e = getenv("MSM_HIP_COMBINE_THREADS")) nthreads = atoi(e) - 1;  // total threads incl. the caller; 1 = serial
    if (nthreads <= 0) return false;
    try {
      for (int i = 0; i < nthreads; i++) threads_.emplace_back([this] { loop(); });
    } catch (...) {
      return !threads_.empty();
    }
    return true;
  }
  void work() {  // all job state is read and written under mu_ (a handful of items of ~50 us each: no contention to speak of)
    std::unique_lock<std::mutex> lk(mu_);
    while (next_ < total_) {
      const int i = next_++;
      const std::function<void(int)>* fn = &fn_;
      lk.unlock();
      (*fn)(i);  // fn_ stays in place until every item of the job has finished
      lk.lock();
      if (++finished_ == total_) cv_done_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::thread> threads_;
  std::function<void(int)> fn_;
  int next_ = 0, total_ = 0, finished_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

inline CombinePool& combine_pool() {
  static CombinePool pool;
  return pool;
}

}  // namespace
