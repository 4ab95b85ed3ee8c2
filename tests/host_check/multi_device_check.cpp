// GPU test program for the in-process "one host thread per GPU" form: one vilo_ctx per visible device, each created and
// used from a thread of its own, disjoint windows per thread (seed 100 + 16 d + i), all threads solving at the same time. Then every window
// is solved again, one after the other, on a fresh context of device 0 from the main thread: a window's result must not depend on which
// device / thread / neighbours it had — compared bitwise. With one visible device the threads share it (two contexts = two streams on
// device 0), which still exercises the concurrent path; the multi-device leg reports itself skipped.
// Prints: devices N / threads T / per window "win d i cost C" / "bitwise_equal 1" / "multi_device_leg ran|skipped".
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/vilo_gpu.h"
#include "../../include/vilo_synth.h"

namespace {
struct Win {
  std::vector<int32_t> lm_start, lm_off, soff;
  std::vector<double> obs, lin, pose, sb, lb, ex, td, lam, tpose, tsb, tlb, tlam, x0, J0, r0;
  std::vector<uint8_t> stereo;
  std::vector<vilo_sample> samples;
  std::vector<vilo_preint> pre;
  vilo_prior prior;
  int n_obs = 0, L = 0;
  void make(const vilo_config &cfg, int seed) {
    vilo_synth_params sp;
    vilo_synth_default_params(&sp, 2);
    sp.n_landmarks = 60; sp.seed = seed;
    int32_t no, ns;
    vilo_synth_sizes(&sp, &no, &ns);
    n_obs = no; L = sp.n_landmarks;
    const int F = VILO_MAX_FRAMES;
    lm_start.resize(L); lm_off.resize(L + 1); soff.resize(F);
    obs.resize(11 * (size_t)no); lin.resize(10 * (F - 1)); pose.resize(7 * F); sb.resize(9 * F); lb.resize(4 * F); ex.resize(14); td.resize(1); lam.resize(L);
    tpose.resize(7 * F); tsb.resize(9 * F); tlb.resize(4 * F); tlam.resize(L); x0.resize(280); J0.resize(96 * 96); r0.resize(96);
    stereo.resize(no); samples.resize(ns); pre.resize(F - 1);
    prior.x0 = x0.data(); prior.J0 = J0.data(); prior.r0 = r0.data();
    vilo_synth_out so = {lm_start.data(), lm_off.data(), obs.data(), stereo.data(), samples.data(), soff.data(), lin.data(),
                         pose.data(), sb.data(), lb.data(), ex.data(), td.data(), lam.data(), tpose.data(), tsb.data(), tlb.data(), tlam.data(), &prior};
    vilo_synth_window(&cfg, &sp, &so);
  }
  vilo_window_desc desc() {
    vilo_window_desc d;
    memset(&d, 0, sizeof(d));
    d.n_frames = VILO_MAX_FRAMES; d.n_landmarks = L; d.n_obs = n_obs; d.use_leg = 1;
    d.lm_start_frame = lm_start.data(); d.lm_obs_offset = lm_off.data(); d.obs = obs.data(); d.obs_is_stereo = stereo.data();
    d.preint = pre.data(); d.preint_imu = nullptr; d.prior = &prior; d.leg_bias_const = 0; d.ex_const = 0; d.td_const = 1;
    return d;
  }
  vilo_window_state state() { return vilo_window_state{pose.data(), sb.data(), lb.data(), ex.data(), td.data(), lam.data()}; }
};

// preintegrate + solve `wins` as one batch on ctx; 0 on success
int run(vilo_ctx *ctx, std::vector<Win> &wins, std::vector<vilo_solve_summary> &sums) {
  const int F = VILO_MAX_FRAMES;
  std::vector<vilo_window_desc> d;
  std::vector<vilo_window_state> s;
  for (Win &w : wins) {
    if (vilo_preintegrate(ctx, F - 1, w.samples.data(), w.soff.data(), w.lin.data(), w.pre.data()) != 0) return 3;
    d.push_back(w.desc()); s.push_back(w.state());
  }
  vilo_solve_opts o;
  vilo_default_solve_opts(&o);
  o.fixed_iterations = 1; o.max_num_iterations = 6;
  sums.resize(wins.size());
  return vilo_solve_windows(ctx, (int)wins.size(), d.data(), s.data(), &o, sums.data());
}
}  // namespace

int main() {
  vilo_config cfg;
  vilo_default_config(&cfg);
  const int ndev = vilo_device_count();
  if (ndev <= 0) { fprintf(stderr, "no GPU\n"); return 2; }
  const int T = ndev >= 2 ? (ndev > 8 ? 8 : ndev) : 2, PER = 24;   // (24 windows per call: the packing and the staged record copy go through the worker pool, from all threads at once)
  printf("devices %d threads %d\n", ndev, T);
  std::vector<std::vector<Win>> wins(T, std::vector<Win>(PER)), again(T, std::vector<Win>(PER));
  for (int d = 0; d < T; ++d)
    for (int i = 0; i < PER; ++i) { wins[d][i].make(cfg, 100 + 16 * d + i); again[d][i].make(cfg, 100 + 16 * d + i); }
  std::vector<std::vector<vilo_solve_summary>> sums(T), sums2(T);
  std::vector<int> rc(T, -1);
  std::vector<std::thread> th;
  for (int d = 0; d < T; ++d)
    th.emplace_back([&, d] {
      vilo_ctx *ctx = nullptr;
      if (vilo_create(&ctx, &cfg, ndev >= 2 ? d : 0) != 0) { rc[d] = 10; return; }   // created AND used on this thread
      int r = 0;
      for (int rep = 0; rep < 3 && r == 0; ++rep) {   // (a few rounds, so that the threads really overlap)
        for (int i = 0; i < PER; ++i) wins[d][i].make(cfg, 100 + 16 * d + i);
        r = run(ctx, wins[d], sums[d]);
      }
      rc[d] = r;
      vilo_destroy(ctx);
    });
  for (auto &t : th) t.join();
  for (int d = 0; d < T; ++d)
    if (rc[d] != 0) { fprintf(stderr, "thread %d rc %d\n", d, rc[d]); return 4; }
  vilo_ctx *c0 = nullptr;
  if (vilo_create(&c0, &cfg, 0) != 0) return 5;
  int equal = 1;
  for (int d = 0; d < T; ++d) {
    if (run(c0, again[d], sums2[d]) != 0) return 6;
    for (int i = 0; i < PER; ++i) {
      printf("win %d %d cost %.17g iterations %d\n", d, i, sums[d][i].final_cost, sums[d][i].iterations);
      Win &a = wins[d][i], &b = again[d][i];
      if (sums[d][i].final_cost != sums2[d][i].final_cost || memcmp(a.pose.data(), b.pose.data(), 8 * a.pose.size()) || memcmp(a.sb.data(), b.sb.data(), 8 * a.sb.size()) ||
          memcmp(a.lb.data(), b.lb.data(), 8 * a.lb.size()) || memcmp(a.lam.data(), b.lam.data(), 8 * a.lam.size()))
        equal = 0;
    }
  }
  vilo_destroy(c0);
  printf("bitwise_equal %d\n", equal);
  printf("multi_device_leg %s\n", ndev >= 2 ? "ran" : "skipped (one visible device: the threads shared it)");
  return equal ? 0 : 7;
}
