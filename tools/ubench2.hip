// Round-2 single-wave micro-benchmarks (gfx950): does an FP64 MFMA overlap with FP64 VALU work of the same wave, and what does the
// 4-block 4x4x4 FP64 MFMA cost / how are its operands laid out?   hipcc --offload-arch=gfx950 -O3 -o ubench2 tools/ubench2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ void k_mix(double *out, long long *clk, double x, int slot) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 0.001, b = x;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * x + 0.5;
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * x + 0.5;
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * x + 0.5;
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * x + 0.5;
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s;
  if (threadIdx.x == 0) clk[slot] = t1 - t0;   // 1024 MFMAs + 1024 NV FMAs
}
__global__ void k_m444(double *out, long long *clk, double x) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double a = threadIdx.x * 0.001, b = x;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0) clk[6] = t1 - t0;
}
// layout probe: D = A B with A[lane] = lane code, B = indicator of one lane at a time
__global__ void k_m444_layout(double *out) {
  // out[probe * 64 + lane]: probe p sets b = 1 on lane p only (a = 100 + lane on all lanes)
  for (int p = 0; p < 64; ++p) {
    double a = 100.0 + threadIdx.x, b = (threadIdx.x == p) ? 1.0 : 0.0;
    double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[p * 64 + threadIdx.x] = c;
  }
}
int main() {
  double *out; long long *clk;
  hipMalloc(&out, 8 * 64 * 64); hipMalloc(&clk, 8 * 16);
  for (int rep = 0; rep < 2; ++rep) {
    k_mix<0><<<1, 64>>>(out, clk, 1.0001, 0);
    k_mix<4><<<1, 64>>>(out, clk, 1.0001, 1);
    k_mix<8><<<1, 64>>>(out, clk, 1.0001, 2);
    k_mix<12><<<1, 64>>>(out, clk, 1.0001, 3);
    k_mix<16><<<1, 64>>>(out, clk, 1.0001, 4);
    k_mix<24><<<1, 64>>>(out, clk, 1.0001, 5);
    k_m444<<<1, 64>>>(out, clk, 1.0001);
  }
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, clk, 128, hipMemcpyDeviceToHost);
  const int nv[6] = {0, 4, 8, 12, 16, 24};
  for (int i = 0; i < 6; ++i) printf("1 mfma f64 16x16x4 + %2d dependent-free fp64 fma: %.1f cyc per group\n", nv[i], h[i] / 1024.0);
  printf("mfma f64 4x4x4 (4 blocks): %.2f cyc/inst\n", h[6] / 1024.0);
  k_m444_layout<<<1, 64>>>(out);
  static double ho[64 * 64]; hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  printf("4x4x4 layout: for B-lane p, lanes with non-zero D and the A-lane they saw (D = A[lane'] * 1):\n");
  for (int p = 0; p < 64; p += 1) {
    printf("  b-lane %2d ->", p);
    for (int l = 0; l < 64; ++l) if (ho[p * 64 + l] != 0.0) printf(" d%d=a%d", l, (int)(ho[p * 64 + l] - 100.0 + 0.5));
    printf("\n");
    if (p == 19) p = 47;
  }
  return 0;
}
