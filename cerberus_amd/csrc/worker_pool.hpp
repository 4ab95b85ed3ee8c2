// A persistent pool of host threads for per-window / per-robot work (packing a batch, a fleet's bookkeeping): the threads are started
// once and parked on a condition variable between jobs — starting and joining 16 threads per call cost ~ 0.3 ms, several times per image
// step of a fleet. Items are handed out one by one through an atomic counter; the caller works too. One job at a time per pool; every
// shared library that includes this has its own pool.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vilo {

class WorkerPool {
 public:
  explicit WorkerPool(int nt) {
    for (int t = 0; t < nt; ++t) th_.emplace_back([this] { worker(); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_work_.notify_all();
    for (auto &x : th_) x.join();
  }
  int size() const { return (int)th_.size(); }
  void run(int n, const std::function<void(int)> &fn) {
    std::lock_guard<std::mutex> serial(run_m_);   // one job at a time
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; n_ = n; next_.store(0); busy_ = (int)th_.size(); ++gen_;
    }
    cv_work_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return busy_ == 0; });
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (int i; (i = next_.fetch_add(1)) < n_;) (*fn_)(i);
  }
  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      drain();
      std::lock_guard<std::mutex> lk(m_);
      if (--busy_ == 0) cv_done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, run_m_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)> *fn_ = nullptr;
  int n_ = 0, busy_ = 0;
  std::atomic<int> next_{0};
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// fn(0) .. fn(n - 1) on up to 16 threads (the caller included), inline when there are fewer than 2 * min_per_thread items
inline void parallel_items(int n, int min_per_thread, const std::function<void(int)> &fn) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int nt = std::max(1, std::min({hw > 0 ? hw : 1, 16, n / std::max(1, min_per_thread)}));
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  static WorkerPool pool(std::max(1, std::min(hw > 0 ? hw : 1, 16) - 1));   // (+ the caller)
  pool.run(n, fn);
}

}  // namespace vilo
