#include <hip/hip_runtime.h>
#include <cstdio>
// single-wave microbenchmarks timed with the shader clock
__global__ void k_fma(double *out, long long *clk, double x) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], x, 1.0);
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) clk[0] = t1 - t0;   // 8192 FMAs
}
__global__ void k_lds128(double *out, long long *clk) {
  __shared__ __attribute__((aligned(16))) double X[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) X[e] = e;
  __syncthreads();
  double2 acc = {0, 0};
  const double2 *p = reinterpret_cast<const double2 *>(X) + (threadIdx.x & 15);
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[u * 16 + (it & 7) * 256];
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
  }
  long long t1 = clock64();
  out[threadIdx.x] = acc.x + acc.y;
  if (threadIdx.x == 0) clk[1] = t1 - t0;   // 4096 ds_read_b128 + 8192 adds
}
__global__ void k_lds64(double *out, long long *clk) {
  __shared__ double X[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) X[e] = e;
  __syncthreads();
  double acc = 0, acc2 = 0;
  const double *p = X + (threadIdx.x & 31);
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[u * 33 + (it & 7) * 512];
#pragma unroll
    for (int u = 0; u < 16; u += 2) { acc += v[u]; acc2 += v[u + 1]; }
  }
  long long t1 = clock64();
  out[threadIdx.x] = acc + acc2;
  if (threadIdx.x == 0) clk[2] = t1 - t0;   // 4096 ds_read_b64 + 4096 adds
}
__global__ void k_div(double *out, long long *clk, double x) {
  double a[4];
  for (int i = 0; i < 4; ++i) a[i] = threadIdx.x + i + 1.5;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = x / a[i] + 1.0;
  }
  long long t1 = clock64();
  out[threadIdx.x] = a[0] + a[1] + a[2] + a[3];
  if (threadIdx.x == 0) clk[3] = t1 - t0;   // 1024 divisions
}
__global__ void k_atomic(double *buf, long long *clk) {
  long long t0 = clock64();
  for (int it = 0; it < 64; ++it) atomicAdd(&buf[(threadIdx.x * 7 + it * 64) & 4095], 1.0);
  __threadfence();
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[4] = t1 - t0;   // 64 wave-wide global f64 atomics
}
__global__ void k_mfma(double *out, long long *clk, double x) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 0.001, b = x;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) clk[5] = t1 - t0;   // 1024 MFMAs
}
int main() {
  double *out, *buf; long long *clk;
  hipMalloc(&out, 8 * 64); hipMalloc(&buf, 8 * 4096 * 64); hipMalloc(&clk, 8 * 8);
  hipMemset(buf, 0, 8 * 4096 * 64);
  for (int rep = 0; rep < 2; ++rep) {
    k_fma<<<1, 64>>>(out, clk, 0.999);
    k_lds128<<<1, 64>>>(out, clk);
    k_lds64<<<1, 64>>>(out, clk);
    k_div<<<1, 64>>>(out, clk, 3.0);
    k_atomic<<<1, 64>>>(buf, clk);
    k_mfma<<<1, 64>>>(out, clk, 1.0001);
  }
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
  printf("fp64 fma: %.2f cyc/inst\n", h[0] / 8192.0);
  printf("ds_read_b128 (+2 adds): %.2f cyc/read\n", h[1] / 4096.0);
  printf("ds_read_b64 (+1 add): %.2f cyc/read\n", h[2] / 4096.0);
  printf("fp64 div (+add): %.2f cyc each\n", h[3] / 1024.0);
  printf("global f64 atomic add: %.1f cyc per wave-instr (1 wave)\n", h[4] / 64.0);
  printf("mfma f64 16x16x4: %.2f cyc/inst\n", h[5] / 1024.0);
  // atomic throughput: whole chip
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k_atomic<<<4096, 64>>>(buf, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("4096 waves x 64 x 64 atomics: %.3f ms -> %.1f G atomics/s\n", ms, 4096.0 * 64 * 64 / ms / 1e6);
  return 0;
}
