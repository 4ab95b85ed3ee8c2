// tests/test_worker_pool.py compiles and runs this: the persistent host pool (cerberus_amd/csrc/worker_pool.hpp) must survive what the
// per-call spawn / join it replaced survived — an exception thrown by an item, a fork()ed child calling into it, process exit.
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <stdexcept>

#include "../../cerberus_amd/csrc/worker_pool.hpp"

int main() {
  std::atomic<long> sum{0};
  vilo::parallel_items(1000, 8, [&](int i) { sum += i; });
  if (sum.load() != 499500) return 1;
  bool caught = false;
  try {
    vilo::parallel_items(1000, 8, [&](int i) { if (i % 97 == 3) throw std::runtime_error("boom"); });
  } catch (const std::runtime_error &) { caught = true; }
  if (!caught) return 2;
  sum = 0;
  vilo::parallel_items(1000, 8, [&](int) { sum += 1; });   // the pool still works after the exception
  if (sum.load() != 1000) return 3;
  fflush(stdout);
  const pid_t p = fork();
  if (p == 0) {   // the child has the pool object and none of its threads
    std::atomic<long> s2{0};
    vilo::parallel_items(1000, 8, [&](int i) { s2 += i; });
    _exit(s2.load() == 499500 ? 0 : 9);
  }
  int st = 0;
  waitpid(p, &st, 0);
  if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) return 4;
  // pools that belong to an object with a lifetime (a context's pipeline lane): jobs of different pools run side by side from their own
  // threads, a pool that was shut down keeps answering (on the caller alone), and its workers are gone
  {
    vilo::WorkerPool a(3), b(3);
    std::atomic<long> sa{0}, sb{0};
    std::thread ta([&] { for (int r = 0; r < 50; ++r) vilo::parallel_items(400, 8, [&](int i) { sa += i; }, &a); });
    std::thread tb([&] { for (int r = 0; r < 50; ++r) vilo::parallel_items(400, 8, [&](int i) { sb += 2 * i; }, &b); });
    ta.join(); tb.join();
    if (sa.load() != 50L * 79800 || sb.load() != 100L * 79800) return 5;
    a.shutdown();
    sa = 0;
    vilo::parallel_items(400, 8, [&](int i) { sa += i; }, &a);
    if (sa.load() != 79800) return 6;
    b.shutdown();
    a.shutdown();   // (twice is harmless)
  }
  {
    // threads of this process: the shared pool's only (the owned pools' workers have left) — /proc/self/task counts them
    usleep(100 * 1000);
    int n = 0;
    char cmd[96];
    snprintf(cmd, sizeof cmd, "ls /proc/%d/task | wc -l", (int)getpid());
    if (FILE *f = popen(cmd, "r")) { if (fscanf(f, "%d", &n) != 1) n = 0; pclose(f); }
    if (n < 1) return 8;
    const int hw = (int)std::thread::hardware_concurrency();
    const int shared = std::max(1, std::min(hw > 0 ? hw : 1, 16) - 1);
    if (n > shared + 3) { printf("threads left: %d (shared pool %d)\n", n, shared); return 7; }
  }
  printf("worker pool OK\n");
  return 0;   // static destruction must not hang or crash either
}
