// Internal declarations shared by the .hip translation units of libvilo_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vilo_gpu.h"
#include "factors.hpp"
#include "worker_pool.hpp"

#define VILO_F 11          // frames in a full window
#define VILO_NB 13         // speed-bias (9) + leg-bias (4) local dims per frame: the "B part"
#define VILO_NP 80         // pose part: 11*6 poses + 6 ex0 + 6 ex1 + 1 td = 79, padded to 80
#define VILO_NPU 79
#define VILO_NCAM (VILO_F * 19 + 13)   // 222
#define VILO_NKERNEL 15
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
  hipEvent_t rec_ev[2] = {nullptr, nullptr};   // "the DMA engine is done with staging chunk i" of vilo_batch_create's record upload
  double last_create_ms[4] = {0, 0, 0, 0}, last_create_bytes = 0.0, last_download_ms = 0.0;   // host wall time of the last vilo_batch_create / _download (vilo_last_create_ms)
  double last_marg_ms = 0.0;       // GPU time of the last vilo_marginalize (linearisation + marginalisation kernels)
  int marg_general_count = 0;      // windows of the last vilo_marginalize that took the global-memory eigen path
  std::string err;
  vilo_config *d_cfg;
  // per-kernel HIP-event timing of the solve pipeline (vilo_set_profiling)
  int profile;
  std::vector<hipEvent_t> pev;      // event pool
  std::vector<int> pev_kind;        // kernel kind of interval i = [pev[2i], pev[2i+1]]
  hipEvent_t prep_ev[2] = {nullptr, nullptr};   // around vilo_batch_prepare's launch (kind 11), read with the next solve's intervals
  bool prep_pending = false;
  double kernel_ms[VILO_NKERNEL];
  long long kernel_launches[VILO_NKERNEL];
  double initial_mu = 1e-8;         // DoglegStrategy's mu at the start of a solve (Ceres: min_mu; vilo_debug_set_initial_mu: per-step comparisons with the oracle)
  int prior_form = 0;               // vilo_set_prior_form: 0 J0 = sqrt(S) V^T as the reference writes it, 1 any X^T with X X^T = A' (pivoted Cholesky factor) where no eigenvalue would be dropped
  int sqrt_info_mode = 0;           // 0: Cholesky of the index-reversed covariance + triangular inverse; 1: the reference's inverse() + LLT, literally
  bool wave_attr_set = false, mid_attr_set = false, mw8_attr_set = false, asm_s_attr_set = false, marg_attr_set = false, prior_attr_set = false;   // dynamic-LDS opt-ins done on this context's device
  // solver form of the batches this context solves (vilo_set_solver_form; -1: chosen from the batch size) and whether batches created on it
  // may use the compact 16-column visual rows (vilo_set_compact_rows). VILO_SOLVER / VILO_NO_COMPACT give the defaults at vilo_create.
  int solver_form = -1;
  int compact_rows = 1;
  // vilo_solve_windows on many host windows: sub-batches of pipe_sub windows through pipe_lanes internal contexts of the same device
  // (created at the first such call, destroyed with this one), one host thread each (vilo_set_host_pipeline; VILO_HOST_PIPELINE=lanes,sub)
  int pipe_lanes = 4, pipe_sub = 1024;
  std::vector<vilo_ctx *> lanes;
  int regime_full = 0;   // a lane: its batches take the kernel set of a full batch whatever their size (BatchDev::full_regime)
  vilo::WorkerPool *pool = nullptr;   // a lane's own host threads (null: the library's shared pool)
  std::mutex dma_m, *dma_turn = nullptr;   // the lanes' uploads take turns (a lane points at its parent's mutex)
};

// grow-only host buffer number `slot` of a context, at least `bytes` long (contents unspecified). Page-locked up to 1 GiB per buffer —
// what the DMA engines read at PCIe rate (pageable memory goes through the runtime's bounce buffer at a fifth of it); beyond that plain
// memory: pinning gigabytes per context is not this library's decision to take. second = size with the low bit telling which kind.
inline void *vilo_host_stage(vilo_ctx *ctx, int slot, size_t bytes) {
  if ((int)ctx->host_stage.size() <= slot) ctx->host_stage.resize(slot + 1, {nullptr, 0});
  auto &hs = ctx->host_stage[slot];
  if ((hs.second & ~(size_t)1) < bytes) {
    if (hs.first) { if (hs.second & 1) (void)hipHostFree(hs.first); else free(hs.first); }
    size_t want = (bytes + bytes / 4 + 1) & ~(size_t)1;
    void *q = nullptr;
    static const bool no_pin = getenv("VILO_NO_PINNED_STAGING") != nullptr;
    if (!no_pin && want <= ((size_t)1 << 30) && hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess && q) { hs.first = q; hs.second = want | 1; }
    else { (void)hipGetLastError(); hs.first = malloc(want); hs.second = hs.first ? want : 0; }
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
int vilo_launch_prepare_preint(vilo_ctx *ctx, int n, const vilo_preint *d_pre, PreintPrepared *d_out, int *d_status, const unsigned char *d_skip, int per_record);
int vilo_launch_prepare_preint_imu(vilo_ctx *ctx, int n, const vilo_preint_imu *d_pre, PreintPrepared *d_out, int *d_status, const unsigned char *d_skip, int per_record);
struct BatchDev;
int vilo_launch_embed_sqrt15(vilo_ctx *ctx, BatchDev &b);
int vilo_repropagate_launch(vilo_ctx *ctx, BatchDev &b, int mode, int stage);

// ---- device-resident hand-over objects (include/vilo_gpu.h) ----
struct vilo_prior_pool {
  int n, device;
  double *dJ, *dr;                 // [n][96*96] (n x n packed, ld n), [n][96]
  std::vector<vilo_prior> meta;    // host mirror: n, blocks, x0 -> x0_store; J0 = r0 = nullptr (device only)
  std::vector<double> x0_store;    // [n][7 * VILO_MAX_PRIOR_BLOCKS]
};
struct PreintStream;
struct PreintImuStream;
struct vilo_preint_streams {
  int n, device, kind;   // kind 0: IMULegIntegrationBase objects (d), 1: IntegrationBase objects (di)
  PreintStream *d;
  PreintImuStream *di;
  // what a reset / push call hands the device (ids, offsets, samples; the push kernel's leg-term scratch): one grow-only device buffer and
  // its page-locked host mirror per kind of call [0 reset, 1 push], and the event after which both may be written again. The calls are
  // asynchronous on the context's stream — a hipMalloc / hipFree per argument and a stream synchronisation per call cost a one-robot
  // replay more than the push kernel itself.
  struct Scratch { void *dev = nullptr, *host = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; bool pending = false; } scr[2];
};
// the prior a window sees: the host struct of its desc, or the host mirror of its pool slot
inline const vilo_prior *vilo_win_prior(const vilo_window_desc &d, const vilo_resident_refs *r) {
  if (r && r->prior_pool) return (r->prior_slot >= 0 && r->prior_slot < r->prior_pool->n) ? &r->prior_pool->meta[r->prior_slot] : nullptr;
  return d.prior;
}
inline double vilo_win_sum_dt(const vilo_window_desc &d, const vilo_resident_refs *r, int k) {
  if (r && r->preint_pool) return r->preint_sum_dt[k];
  return d.use_leg ? d.preint[k].sum_dt : d.preint_imu[k].sum_dt;
}
struct vilo_batch;
// vilo_batch.hip: a call on many host windows cut into sub-batches over the context's pipeline lanes (false: not a call to cut)
bool vilo_run_on_lanes(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, vilo_window_state *inout,
                       const std::function<int(vilo_ctx *lane, int w0, int n)> &fn, int *rc_out);
int vilo_batch_create_refs(vilo_ctx *ctx, int W, const vilo_window_desc *in, const vilo_resident_refs *refs, const vilo_window_state *init, vilo_batch **out);
