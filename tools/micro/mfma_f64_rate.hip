// Issue rate of v_mfma_f64_16x16x4_f64 from ONE wave versus several waves of a SIMD (gfx950). Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip
// Each wave issues N independent MFMAs (15 accumulators round-robin, like the pose-system tiles of the solver) and reports cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double *out, long long *cyc, int iters) {
  d4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, sizeof(double) * 256 * 4096); hipMalloc(&cyc, sizeof(long long) * 4 * 4096);
  const int iters = 200;
  for (int waves_per_cu : {1, 4, 8, 16}) {   // 1: one wave on one SIMD of the CU; 4: one per SIMD; 8: two per SIMD; 16: four per SIMD
    const int threads = waves_per_cu >= 4 ? 256 : 64, blocks_per_cu = waves_per_cu >= 4 ? waves_per_cu / 4 : 1;
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<15>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[64]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 16; ++i) mean += h[i]; mean /= 16;
    printf("waves per CU %2d (%d per SIMD): %.1f cycles per MFMA per wave, %.1f per SIMD\n", waves_per_cu, waves_per_cu >= 4 ? waves_per_cu / 4 : 1,
           mean / (iters * 15.0), mean / (iters * 15.0) / (waves_per_cu >= 4 ? waves_per_cu / 4 : 1));
  }
  return 0;
}
