// Operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment: one-hot A (lane la) x one-hot B (lane lb) -> which lanes
// of D are 1. Prints, per lane, (block, row, k) of A, (block, k, column) of B and (block, row, column) of D.
// Build: hipcc --offload-arch=gfx950 -O3 -o l mfma_f64_4x4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned long long *out) {   // out[la * 64 + lb] = bitmask of D lanes that are 1
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m;
    }
}
int main() {
  unsigned long long *d; (void)hipMalloc(&d, 8 * 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  std::vector<unsigned long long> h(4096);
  (void)hipMemcpy(h.data(), d, 8 * 4096, hipMemcpyDeviceToHost);
  // hypothesis: A lane = 16 * blk + 4 * k + i ?  print for la in the first block which lb give output and where
  for (int la : {0, 1, 4, 5, 16, 21}) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb]) {
        unsigned long long m = h[la * 64 + lb];
        printf("  [B %d -> D", lb);
        for (int l = 0; l < 64; ++l) if (m >> l & 1) printf(" %d", l);
        printf("]");
      }
    printf("\n");
  }
  // full check of the hypothesis A: lane = 16 b + 4 k + i, B: lane = 16 b + 4 k + j, D: lane = 16 b + 4 j + i  (and alternatives)
  const char *names[4] = {"A=16b+4k+i B=16b+4k+j D=16b+4j+i", "A=16b+4k+i B=16b+4k+j D=16b+4i+j", "A=16b+4i+k B=16b+4j+k D=16b+4j+i", "A=16b+4i+k B=16b+4j+k D=16b+4i+j"};
  for (int hyp = 0; hyp < 4; ++hyp) {
    int bad = 0;
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb) {
        const int ba = la / 16, bb = lb / 16;
        const int ka = (hyp < 2) ? (la / 4) % 4 : la % 4, ia = (hyp < 2) ? la % 4 : (la / 4) % 4;
        const int kb = (hyp < 2) ? (lb / 4) % 4 : lb % 4, jb = (hyp < 2) ? lb % 4 : (lb / 4) % 4;
        unsigned long long want = 0;
        if (ba == bb && ka == kb) want = 1ull << (16 * ba + ((hyp & 1) ? 4 * ia + jb : 4 * jb + ia));
        if (h[la * 64 + lb] != want) ++bad;
      }
    printf("hypothesis %s: %d mismatches of 4096\n", names[hyp], bad);
  }
  return 0;
}
