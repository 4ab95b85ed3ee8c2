// wave_sum_dpp / wave_max_dpp (cerberus_amd/csrc/solve_common.hpp) against a host sum of the same 64 values:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I cerberus_amd/csrc tools/micro/dpp_reduce_check.hip -o _tmpbin/dpp_reduce_check && _tmpbin/dpp_reduce_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "solve_common.hpp"

__global__ void k(double *out, const double *in) {
  const int l = threadIdx.x, w = blockIdx.x;
  const double v = in[64 * w + l];
  out[128 * w + l] = wave_sum_dpp(v);
  out[128 * w + 64 + l] = wave_max_dpp(fabs(v));
}

int main() {
  const int W = 256;
  std::vector<double> h(64 * W), o(128 * W);
  std::mt19937_64 g(7);
  std::normal_distribution<double> n(0.0, 1.0);
  for (auto &x : h) x = n(g);
  double *di, *d_o;
  hipMalloc(&di, h.size() * 8); hipMalloc(&d_o, o.size() * 8);
  hipMemcpy(di, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, d_o, di);
  hipMemcpy(o.data(), d_o, o.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0.0;
  int bad = 0;
  for (int w = 0; w < W; ++w) {
    long double s = 0.0; double m = 0.0;
    for (int l = 0; l < 64; ++l) { s += h[64 * w + l]; m = std::fmax(m, std::fabs(h[64 * w + l])); }
    for (int l = 0; l < 64; ++l) {
      worst = std::fmax(worst, std::fabs(o[128 * w + l] - (double)s));
      if (o[128 * w + l] != o[128 * w] || o[128 * w + 64 + l] != m) ++bad;
    }
  }
  printf("wave_sum_dpp: max |error| %.3e (64 terms of unit scale), lanes that disagree or wrong maxima: %d -> %s\n", worst, bad, (worst < 1e-13 && bad == 0) ? "OK" : "FAILED");
  return (worst < 1e-13 && bad == 0) ? 0 : 1;
}
