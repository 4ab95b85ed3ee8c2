// Rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per instruction, 256 multiply-adds) beside v_mfma_f64_16x16x4_f64 (1024)
// on gfx950: would the upper 4 x 4 blocks of a symmetric 16 x 16 Gram tile (10 of 16 blocks = three instructions of four blocks) be
// cheaper than the one 16 x 16 x 4 instruction that also computes the lower half? Build: hipcc --offload-arch=gfx950 -O3 -o m mfma_f64_4x4_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, bool SMALL>
__global__ void __launch_bounds__(256) k(double *out, long long *cyc, int iters) {
  d4 acc[NACC];
  double sc[NACC];
  for (int t = 0; t < NACC; ++t) { acc[t] = d4{0, 0, 0, 0}; sc[t] = 0.0; }
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      if (SMALL) sc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3] + sc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC, bool SMALL>
void run(double *out, long long *cyc, int iters) {
  printf("-- %s, %d independent accumulators per wave\n", SMALL ? "v_mfma_f64_4x4x4_4b (256 MAC)" : "v_mfma_f64_16x16x4 (1024 MAC)", NACC);
  for (int waves_per_cu : {0, 4, 8}) {   // 0: ONE wave on the whole chip; 4: one per SIMD on every CU; 8: two per SIMD
    const int threads = waves_per_cu >= 4 ? 256 : 64, blocks_per_cu = waves_per_cu >= 4 ? waves_per_cu / 4 : 1;
    const int blocks = waves_per_cu == 0 ? 1 : 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; const int nw = blocks * (threads / 64) < 16 ? blocks * (threads / 64) : 16; for (int i = 0; i < nw; ++i) mean += h[i]; mean /= nw;
    const double n = (double)blocks * (threads / 64) * iters * NACC, mac = SMALL ? 256.0 : 1024.0;
    printf("waves per CU %2d: %7.1f clock64 ticks per instruction per wave; kernel %.3f ms -> %.1f TFLOP/s, %.2f MAC per tick per wave\n", waves_per_cu,
           mean / (iters * (double)NACC), ms, n * 2 * mac / (ms * 1e-3) / 1e12, mac / (mean / (iters * (double)NACC)));
  }
}
int main() {
  double *out; long long *cyc;
  (void)hipMalloc(&out, sizeof(double) * 256 * 4096); (void)hipMalloc(&cyc, sizeof(long long) * 4 * 4096);
  run<1, false>(out, cyc, 4000); run<1, true>(out, cyc, 4000);     // dependent chains (one accumulator)
  run<8, false>(out, cyc, 1000); run<8, true>(out, cyc, 1000);     // independent accumulators
  return 0;
}
