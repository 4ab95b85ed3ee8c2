// A 16 x 16 x 4 FP64 tile update as FOUR v_mfma_f64_4x4x4_4b: block s of instruction rho computes rows 4 rho .. 4 rho + 3, columns 4 s .. 4 s + 3 —
// lane 16 i + 4 s + j, exactly where register rho of the 16 x 16 x 4 accumulator keeps that element — if its A operand carries row block rho in
// all four slots (lane (k, s, i) <- lane (k, rho, i) of the ordinary operand: ds_swizzle_b32, bit mode, and 0x13 or rho << 2) and its B operand is
// the ordinary one. Checks the result against v_mfma_f64_16x16x4 (bitwise?) and times the landmark-Schur k-step of the pose system both ways:
// 5 operands, 15 lower tiles, one wave per SIMD on the whole chip. Build: hipcc --offload-arch=gfx950 -O3 -o t mfma_f64_tile_via_4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int RHO>
__device__ __forceinline__ double rep_rows(double a) {
  constexpr int pat = 0x13 | (RHO << 2) << 5;   // and_mask 0b10011, or_mask rho << 2, xor_mask 0
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(a), pat), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(a), pat);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ d4 tile_via4(double a, double b, d4 c) {
  c[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(rep_rows<0>(a), b, c[0], 0, 0, 0);
  c[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(rep_rows<1>(a), b, c[1], 0, 0, 0);
  c[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(rep_rows<2>(a), b, c[2], 0, 0, 0);
  c[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(rep_rows<3>(a), b, c[3], 0, 0, 0);
  return c;
}
__global__ void k_check(const double *a, const double *b, double *o16, double *o4) {
  const int l = threadIdx.x;
  d4 c16 = {0, 0, 0, 0}, c4 = {0, 0, 0, 0};
  for (int s = 0; s < 8; ++s) {
    c16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[64 * s + l], b[64 * s + l], c16, 0, 0, 0);
    c4 = tile_via4(a[64 * s + l], b[64 * s + l], c4);
  }
  for (int r = 0; r < 4; ++r) { o16[64 * r + l] = c16[r]; o4[64 * r + l] = c4[r]; }
}
__device__ constexpr int tI[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4}, tJ[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
template <bool SMALL>
__global__ void __launch_bounds__(64) k_schur(double *out, const double *w, int ksteps) {
  d4 acc[15];
  for (int t = 0; t < 15; ++t) acc[t] = d4{0, 0, 0, 0};
  const int l = threadIdx.x;
  for (int ks = 0; ks < ksteps; ++ks) {
    double op[5];
    for (int X = 0; X < 5; ++X) op[X] = w[(size_t)(ks * 5 + X) * 64 + l];
    const double ei = 0.5 + 1e-3 * (ks & 7);
    if (SMALL) {
#pragma unroll
      for (int I = 0; I < 5; ++I) {
        const double a = -(op[I] * ei);
        const double a0 = rep_rows<0>(a), a1 = rep_rows<1>(a), a2 = rep_rows<2>(a), a3 = rep_rows<3>(a);
#pragma unroll
        for (int J = 0; J <= I; ++J) {
          const int t = I * (I + 1) / 2 + J;
          acc[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, op[J], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, op[J], acc[t][1], 0, 0, 0);
          acc[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, op[J], acc[t][2], 0, 0, 0);
          acc[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, op[J], acc[t][3], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 15; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(op[tI[t]] * ei), op[tJ[t]], acc[t], 0, 0, 0);
    }
  }
  double s = 0;
  for (int t = 0; t < 15; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 64 + l] = s;
}
int main() {
  std::vector<double> ha(512), hb(512);
  for (int i = 0; i < 512; ++i) { ha[i] = 0.37 + 0.011 * ((i * 37) % 101) - 0.5; hb[i] = 1.3 - 0.017 * ((i * 53) % 89); }
  double *a, *b, *o16, *o4, *out, *w;
  (void)hipMalloc(&a, 4096); (void)hipMalloc(&b, 4096); (void)hipMalloc(&o16, 2048); (void)hipMalloc(&o4, 2048);
  (void)hipMemcpy(a, ha.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, a, b, o16, o4);
  std::vector<double> r16(256), r4(256);
  (void)hipMemcpy(r16.data(), o16, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(r4.data(), o4, 2048, hipMemcpyDeviceToHost);
  int neq = 0; double worst = 0;
  for (int i = 0; i < 256; ++i) { if (r16[i] != r4[i]) ++neq; double d = r16[i] - r4[i]; if (d < 0) d = -d; if (d > worst) worst = d; }
  printf("four 4x4x4_4b vs one 16x16x4 over 8 k-steps: %d of 256 entries differ, largest difference %.3e (entries ~ %.3f)\n", neq, worst, r16[5]);
  const int ksteps = 512, blocks = 1024;   // one wave per SIMD on 256 CUs
  (void)hipMalloc(&out, 8 * 64 * blocks); (void)hipMalloc(&w, 8 * 64 * 5 * ksteps);
  (void)hipMemset(w, 0, 8 * 64 * 5 * ksteps);
  for (int rep = 0; rep < 2; ++rep)
    for (int small = 0; small < 2; ++small) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (small) hipLaunchKernelGGL(k_schur<true>, dim3(blocks), dim3(64), 0, 0, out, w, ksteps);
      else hipLaunchKernelGGL(k_schur<false>, dim3(blocks), dim3(64), 0, 0, out, w, ksteps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.3f ms for %d k-steps x 15 tiles on 1024 waves -> %.1f ns per k-step, %.1f TFLOP/s\n", small ? "4 x 4x4x4_4b + ds_swizzle" : "16x16x4            ", ms, ksteps,
             ms * 1e6 / ksteps, (double)blocks * ksteps * 15 * 2048 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
