// A persistent pool of host threads for per-window / per-robot work (packing a batch, a fleet's bookkeeping): the threads are started
// once and parked on a condition variable between jobs — starting and joining 16 threads per call cost ~ 0.3 ms, several times per image
// step of a fleet. Items are handed out one by one through an atomic counter; the caller works too. One job at a time per pool; every
// shared library that includes this has its own pool.
#pragma once
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vilo {

// Three properties the per-call spawn / join it replaced had for free, kept here on purpose:
//   * fork(): a child process inherits the pool object but none of its threads. The pool remembers the pid that started the threads; in any
//     other process run() works through the items on the calling thread (no waiting for workers that do not exist).
//   * exceptions: fn throwing on the CALLING thread must not unwind run() while workers still dereference fn_ — the caller's share runs under
//     try / catch, the remaining items are cancelled, run() waits for busy_ == 0 and only then rethrows. A worker's exception is carried
//     over to the caller the same way (first one wins).
//   * process exit: the workers are detached and the shared state lives in a block that is never freed, so static destruction neither
//     joins (a child after fork() would join threads it never had) nor pulls memory from under a worker that is just waking up.
class WorkerPool {
  struct State {
    std::mutex m, run_m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *fn = nullptr;
    int n = 0, busy = 0;
    std::atomic<int> next{0};
    unsigned long long gen = 0;
    bool stop = false;   // shutdown(): the workers leave (the block itself stays, like everything else here)
    std::exception_ptr err;
  };

 public:
  explicit WorkerPool(int nt) : st_(new State), nt_(nt), pid_(getpid()) {
    for (int t = 0; t < nt; ++t) std::thread([s = st_] { worker(s); }).detach();
  }
  ~WorkerPool() {}   // (the workers stay parked on st_, which is leaked deliberately: see above)
  int size() const { return nt_; }
  // A pool that belongs to an object with a lifetime (a context's pipeline lane): its workers end; run() afterwards works on the caller alone.
  void shutdown() {
    State &s = *st_;
    std::lock_guard<std::mutex> serial(s.run_m);
    { std::lock_guard<std::mutex> lk(s.m); s.stop = true; }
    s.cv_work.notify_all();
    pid_ = -1;
  }
  void run(int n, const std::function<void(int)> &fn) {
    if (getpid() != pid_) {   // a fork()ed child: no workers here
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    State &s = *st_;
    std::lock_guard<std::mutex> serial(s.run_m);   // one job at a time
    {
      std::lock_guard<std::mutex> lk(s.m);
      s.fn = &fn; s.n = n; s.next.store(0); s.busy = nt_; ++s.gen; s.err = nullptr;
    }
    s.cv_work.notify_all();
    drain(&s);
    std::exception_ptr err;
    {
      std::unique_lock<std::mutex> lk(s.m);
      s.cv_done.wait(lk, [&] { return s.busy == 0; });
      s.fn = nullptr;
      err = s.err;
      s.err = nullptr;
    }
    if (err) std::rethrow_exception(err);
  }

 private:
  static void drain(State *s) {
    try {
      for (int i; (i = s->next.fetch_add(1)) < s->n;) (*s->fn)(i);
    } catch (...) {
      s->next.store(s->n);   // cancel what has not been handed out
      std::lock_guard<std::mutex> lk(s->m);
      if (!s->err) s->err = std::current_exception();
    }
  }
  static void worker(State *s) {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(s->m);
        s->cv_work.wait(lk, [&] { return s->gen != seen || s->stop; });
        if (s->stop) return;
        seen = s->gen;
      }
      drain(s);
      std::lock_guard<std::mutex> lk(s->m);
      if (--s->busy == 0) s->cv_done.notify_one();
    }
  }
  State *st_;
  int nt_;
  pid_t pid_;
};

// fn(0) .. fn(n - 1) on up to 16 threads (the caller included), inline when there are fewer than 2 * min_per_thread items; `own`: a pool
// of the caller's instead of the library's shared one (jobs of one pool run one at a time: callers that work side by side bring their own)
inline void parallel_items(int n, int min_per_thread, const std::function<void(int)> &fn, WorkerPool *own = nullptr) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int nt = std::max(1, std::min({hw > 0 ? hw : 1, 16, n / std::max(1, min_per_thread)}));
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  if (own) { own->run(n, fn); return; }
  static WorkerPool pool(std::max(1, std::min(hw > 0 ? hw : 1, 16) - 1));   // (+ the caller)
  pool.run(n, fn);
}

}  // namespace vilo
