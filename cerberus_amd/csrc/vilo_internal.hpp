// Internal declarations shared by the .hip translation units of libvilo_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/vilo_gpu.h"
#include "factors.hpp"

#define VILO_F 11          // frames in a full window
#define VILO_NB 13         // speed-bias (9) + leg-bias (4) local dims per frame: the "B part"
#define VILO_NP 80         // pose part: 11*6 poses + 6 ex0 + 6 ex1 + 1 td = 79, padded to 80
#define VILO_NPU 79
#define VILO_NCAM (VILO_F * 19 + 13)   // 222
#define VILO_GRAM 351      // packed upper triangle of the 26 x 26 per-(group, t) Gram matrix
#define VILO_GCOLS 26      // [pose_s 6 | pose_j 6 | ex0 6 | ex1 6 | td 1 | r 1]

struct vilo_ctx {
  vilo_config cfg;
  int device;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  double last_solve_ms;
  // device memory pool: batches are built and torn down once per image in a replay; their arena chunks come from and return to
  // this free list (grow-only, released by vilo_destroy) instead of ~65 hipMalloc / hipFree per batch
  std::vector<std::pair<void *, size_t>> pool_free;
  // reusable host staging (grow-only): the wave-packed observation image and the prior staging of vilo_batch_create
  std::vector<std::pair<void *, size_t>> host_stage;
  double last_marg_ms = 0.0;       // GPU time of the last vilo_marginalize (linearisation + marginalisation kernels)
  int marg_general_count = 0;      // windows of the last vilo_marginalize that took the global-memory eigen path
  std::string err;
  vilo_config *d_cfg;
  // per-kernel HIP-event timing of the solve pipeline (vilo_set_profiling)
  int profile;
  std::vector<hipEvent_t> pev;      // event pool
  std::vector<int> pev_kind;        // kernel kind of interval i = [pev[2i], pev[2i+1]]
  double kernel_ms[8];
  long long kernel_launches[8];
};
#define VILO_NKERNEL 8

// grow-only host buffer number `slot` of a context, at least `bytes` long (contents unspecified)
inline void *vilo_host_stage(vilo_ctx *ctx, int slot, size_t bytes) {
  if ((int)ctx->host_stage.size() <= slot) ctx->host_stage.resize(slot + 1, {nullptr, 0});
  auto &hs = ctx->host_stage[slot];
  if (hs.second < bytes) {
    free(hs.first);
    hs.second = bytes + bytes / 4;
    hs.first = malloc(hs.second);
  }
  return hs.first;
}


#define VILO_HIP(call)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      char buf_[512];                                                                                    \
      snprintf(buf_, sizeof(buf_), "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      if (ctx) ctx->err = buf_;                                                                          \
      return VILO_ERR_HIP;                                                                               \
    }                                                                                                    \
  } while (0)

// RAII device buffer for the convenience (host-pointer) entry points
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) {
    bytes = n;
    if (n == 0) { p = nullptr; return hipSuccess; }
    return hipMalloc(&p, n);
  }
  template <class T> T *as() { return (T *)p; }
};

// prepared preintegration record consumed by the factor kernels: head (126 doubles) + sqrt_info (31x31)
struct PreintPrepared {
  vilo::PreintHead head;
  double sqrt_info[31 * 31];  // upper triangular, row-major
};
static_assert(sizeof(vilo::PreintHead) == 8 * (33 + 45 + 36 + 12), "PreintHead layout");
static_assert(sizeof(PreintPrepared) == 8 * 1087, "SURVEY 8(d): 1087 doubles per preintegration record");

// kernels_eval.hip
int vilo_launch_prepare_preint(vilo_ctx *ctx, int n, const vilo_preint *d_pre, PreintPrepared *d_out, int *d_status);
int vilo_launch_prepare_preint_imu(vilo_ctx *ctx, int n, const vilo_preint_imu *d_pre, PreintPrepared *d_out, int *d_status);
struct BatchDev;
int vilo_launch_sqrt_transpose(vilo_ctx *ctx, BatchDev &b);
int vilo_launch_embed_sqrt15(vilo_ctx *ctx, BatchDev &b);
