// Which waves of a workgroup share a SIMD's FP64 matrix pipe, and at what rate one wave alone issues v_mfma_f64_16x16x4 — one workgroup
// of up to eight waves on an otherwise idle chip (the situation of k_solve_mw8: kernels_mw8.hip). Output: profiles/round4_mfma_simd_share.txt.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_simd_share tools/micro/mfma_simd_share.hip && ./mfma_simd_share
// mask: bit w = hardware wave w of the workgroup runs the loop; NACC: independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(double *out, long long *clk, int iters, unsigned mask) {
  const int w = threadIdx.x >> 6;
  if (!((mask >> w) & 1)) return;
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[w] = t1 - t0;
}
template <int NACC> void run(unsigned mask) {
  double *out; long long *clk; (void)hipMalloc(&out, 512 * 8); (void)hipMalloc(&clk, 64); (void)hipMemset(clk, 0, 64);
  const int iters = 1000;
  k<NACC><<<1, 512>>>(out, clk, iters, mask); k<NACC><<<1, 512>>>(out, clk, iters, mask);
  (void)hipDeviceSynchronize();
  long long h[8]; (void)hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
  printf("NACC=%d mask=%02x cycles/MFMA per wave:", NACC, mask);
  for (int w = 0; w < 8; ++w) if ((mask >> w) & 1) printf(" w%d %.1f", w, (double)h[w] / (iters * NACC));
  printf("\n"); (void)hipFree(out); (void)hipFree(clk);
}
int main() {
  for (unsigned m : {0x01u, 0x03u, 0x11u, 0x05u, 0x0fu, 0xffu, 0x33u, 0x55u}) { run<1>(m); run<4>(m); run<8>(m); }
  return 0;
}
