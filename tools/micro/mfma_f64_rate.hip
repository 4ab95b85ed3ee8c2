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
template <int NACC>
void run(double *out, long long *cyc, int iters) {
  printf("-- %d independent accumulators per wave\n", NACC);
  for (int waves_per_cu : {0, 1, 4, 8, 16}) {   // 0: ONE wave on the whole chip; 1: one wave per CU; 4: one per SIMD; 8: two per SIMD; 16: four per SIMD
    const int threads = waves_per_cu >= 4 ? 256 : 64, blocks_per_cu = waves_per_cu >= 4 ? waves_per_cu / 4 : 1;
    const int blocks = waves_per_cu == 0 ? 1 : 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);   // warm
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; const int nw = blocks * (threads / 64) < 16 ? blocks * (threads / 64) : 16; for (int i = 0; i < nw; ++i) mean += h[i]; mean /= nw;
    const double mfmas = (double)blocks * (threads / 64) * iters * NACC;
    printf("waves per CU %2d: %7.1f clock64 ticks per MFMA per wave; kernel %.3f ms -> %.1f TFLOP/s (2048 flop per MFMA), %.1f ns per MFMA per wave\n", waves_per_cu,
           mean / (iters * (double)NACC), ms, mfmas * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * (double)NACC));
  }
}
int main() {
  double *out; long long *cyc;
  (void)hipMalloc(&out, sizeof(double) * 256 * 4096); (void)hipMalloc(&cyc, sizeof(long long) * 4 * 4096);
  run<4>(out, cyc, 2000);
  run<15>(out, cyc, 600);
  return 0;
}
