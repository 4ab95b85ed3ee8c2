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
  printf("worker pool OK\n");
  return 0;   // static destruction must not hang or crash either
}
