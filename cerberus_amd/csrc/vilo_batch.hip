// Device-resident batches of independent sliding windows: packing (host -> HBM layout), the solve driver and
// result download. Replaces the pack / unpack halves of Estimator::optimization():
//   vector2double  estimator.cpp:848-901   (para_* arrays, inverse depths in feature-list order)
//   problem build  estimator.cpp:1059-1216 (which residual blocks exist; here: landmark-major chunk tables)
//   double2vector  estimator.cpp:903-1003  (gauge fix: vilo_gauge_fix)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "solver_types.hpp"
#include "worker_pool.hpp"

int vilo_solve_launch(vilo_ctx *ctx, BatchDev &b, const vilo_solve_opts *o);

struct vilo_batch {
  BatchDev d;
  std::vector<std::pair<void *, size_t>> chunks_dev;   // arena chunks (from / back to ctx->pool_free)
  char *cur = nullptr;
  size_t cur_left = 0;
  std::vector<int> lm_off_host;     // per window
  std::vector<int> perm_host;       // device order -> original landmark index (per window, concatenated)
  std::vector<int> L_host;
  int W;
  // the launch sequence of one solve (5 + 7 x max_num_iterations kernels) as a hipGraph, captured when the same resident batch is
  // solved a second time with the same options (vilo_batch_reset + vilo_batch_solve loops: replays, Monte-Carlo seeds, bench)
  hipGraphExec_t gexec = nullptr;
  vilo_solve_opts gopts;
  int g_sqrt_info_mode = 0, g_rp_on = 0, g_solver_form = -1;   // context / batch state the captured launch sequence depends on (part of the cache key)
  double g_initial_mu = 1e-8;
  // re-propagation buffers (vilo_batch_set_samples): reused by later calls while they are large enough (the arena cannot free)
  vilo_sample *rp_s = nullptr; int *rp_o = nullptr; double *rp_t = nullptr; size_t rp_cap = 0;
  double *rp_ff = nullptr;          // [W * 10][VILO_FF_N]
  double *rp_ff0 = nullptr;         // the same right after the objects' first integration: what vilo_batch_reset brings back (contact_sensor_type 2)
  vilo_preint *rp_orig = nullptr;   // [W * 10] the records as created, kept from the first vilo_batch_set_samples on
  int n_solves = 0;
  bool graph_failed = false;
  // what vilo_batch_prepare needs to run the sqrt_info preparation again (the reference does it in every IMULegFactor::Evaluate)
  void *d_pre = nullptr;
  bool leg = true;
  int *d_prep_bad = nullptr;   // [W * 10] covariance of this record not positive definite
};

namespace {

// win_bad[w] = any live interval of window w whose covariance had no sqrt_info (prep_bad, written by the preparation)
// What the solver reads of an IMULegIntegrationBase record (vilo_preint, 1955 doubles): the 33 scalars, the bias columns 21 .. 30 of the
// Jacobian's rows 0 .. 20 (PreintHead, factors.hpp) and ONE triangle of the covariance (k_prepare_preint's Cholesky of the index-reversed
// matrix takes the source's upper triangle; the literal inverse() route reads all of it and keeps the full upload). Host windows bring
// their records up in this form — 739 doubles, 38 % of the bytes — and the device lays them out as the record the kernels index: the
// Jacobian entries no factor reads are zero in that copy, the covariance's lower triangle is the mirror of the upper.
#define REC_C_JAC 33
#define REC_C_TRI (33 + 21 * 10)
#define REC_C_N (REC_C_TRI + 31 * 32 / 2)   // 739
static inline int rec_tri_off(int r) { return r * 31 - r * (r - 1) / 2; }   // first entry of row r (columns r .. 30) of the packed upper triangle
static void rec_compact(const vilo_preint *src, double *dst) {
  memcpy(dst, src, sizeof(double) * 33);
  for (int r = 0; r < 21; ++r) memcpy(dst + REC_C_JAC + 10 * r, src->jacobian + r * 31 + 21, sizeof(double) * 10);
  for (int r = 0; r < 31; ++r) memcpy(dst + REC_C_TRI + rec_tri_off(r), src->covariance + r * 31 + r, sizeof(double) * (31 - r));
}
__global__ void __launch_bounds__(256) k_expand_records(int n, const double *comp, vilo_preint *out) {
  const int f = blockIdx.x;
  if (f >= n) return;
  const double *c = comp + (size_t)f * REC_C_N;
  double *o = (double *)(out + f);
  for (int e = threadIdx.x; e < (int)(sizeof(vilo_preint) / sizeof(double)); e += 256) {
    double v;
    if (e < 33) v = c[e];
    else if (e < 33 + 961) {
      const int r = (e - 33) / 31, q = (e - 33) - 31 * r;
      v = (r < 21 && q >= 21) ? c[REC_C_JAC + 10 * r + q - 21] : 0.0;
    } else {
      const int r = (e - 33 - 961) / 31, q = (e - 33 - 961) - 31 * r;
      const int a = min(r, q), b = max(r, q);
      v = c[REC_C_TRI + a * 31 - a * (a - 1) / 2 + (b - a)];
    }
    o[e] = v;
  }
}

__global__ void k_fold_win_bad(int W, const int *prep_bad, const unsigned char *imu_skip, int *win_bad) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  int bad = 0;
  for (int k = 0; k < 10; ++k) bad |= (!imu_skip[(size_t)w * 10 + k] && prep_bad[(size_t)w * 10 + k]) ? 1 : 0;
  win_bad[w] = bad;
}

// bump allocation out of 64 MB (or larger) arena chunks; chunks are recycled through the context's free list
int dev_alloc_bytes(vilo_ctx *ctx, vilo_batch *bt, void **p, size_t bytes) {
  *p = nullptr;
  bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  if (bt->cur_left < bytes) {
    const size_t want = std::max<size_t>(bytes, (size_t)64 << 20);
    int best = -1;
    for (int i = 0; i < (int)ctx->pool_free.size(); ++i)
      if (ctx->pool_free[i].second >= bytes && (best < 0 || ctx->pool_free[i].second < ctx->pool_free[best].second)) best = i;
    std::pair<void *, size_t> ch;
    if (best >= 0) {
      ch = ctx->pool_free[best];
      ctx->pool_free.erase(ctx->pool_free.begin() + best);
    } else {
      void *q = nullptr;
      VILO_HIP(hipMalloc(&q, want));
      ch = {q, want};
    }
    bt->chunks_dev.push_back(ch);
    bt->cur = (char *)ch.first; bt->cur_left = ch.second;
  }
  *p = bt->cur;
  bt->cur += bytes; bt->cur_left -= bytes;
  return VILO_OK;
}
template <class T>
int dev_alloc(vilo_ctx *ctx, vilo_batch *bt, T **p, size_t n) {
  void *q = nullptr;
  int rc = dev_alloc_bytes(ctx, bt, &q, std::max<size_t>(n, 1) * sizeof(T));
  *p = (T *)q;
  return rc;
}
template <class T>
int dev_upload(vilo_ctx *ctx, vilo_batch *bt, T **p, const std::vector<T> &h) {
  int rc = dev_alloc(ctx, bt, p, h.size());
  if (rc != VILO_OK) return rc;
  if (!h.empty()) VILO_HIP(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return VILO_OK;
}
template <class T>
int dev_upload_raw(vilo_ctx *ctx, vilo_batch *bt, T **p, const T *h, size_t n) {
  int rc = dev_alloc(ctx, bt, p, n);
  if (rc != VILO_OK) return rc;
  if (n) VILO_HIP(hipMemcpy(*p, h, n * sizeof(T), hipMemcpyHostToDevice));
  return VILO_OK;
}
// The tables of a batch are a dozen and a half small arrays: uploaded one by one, a window's batch pays a blocking copy for each (what a
// frame-by-frame caller pays per image). They are laid out in ONE host blob at the offsets of one device allocation and go up in one copy;
// the device pointers are set when the blob is flushed (nothing may read them before). Arrays of 256 KB and more keep their own copy
// straight from the caller's memory.
struct UploadBlob {
  struct Item { void **pp; size_t off; };
  std::vector<char> host;
  std::vector<Item> items;
  template <class T>
  int add(vilo_ctx *ctx, vilo_batch *bt, T **p, const T *h, size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= ((size_t)256 << 10)) return dev_upload_raw(ctx, bt, p, h, n);
    const size_t off = (host.size() + 255) & ~(size_t)255;
    host.resize(off + std::max<size_t>(bytes, sizeof(T)));
    if (bytes) memcpy(host.data() + off, h, bytes);
    *p = nullptr;
    items.push_back({(void **)p, off});
    return VILO_OK;
  }
  template <class T>
  int add(vilo_ctx *ctx, vilo_batch *bt, T **p, const std::vector<T> &h) { return add(ctx, bt, p, h.data(), h.size()); }
  int flush(vilo_ctx *ctx, vilo_batch *bt) {
    if (items.empty()) return VILO_OK;
    void *base = nullptr;
    int rc = dev_alloc_bytes(ctx, bt, &base, host.size());
    if (rc != VILO_OK) return rc;
    VILO_HIP(hipMemcpy(base, host.data(), host.size(), hipMemcpyHostToDevice));
    for (const Item &it : items) *it.pp = (char *)base + it.off;
    items.clear(); host.clear();
    return VILO_OK;
  }
};
#define TRYB(x) do { int rc_ = (x); if (rc_ != VILO_OK) { vilo_batch_destroy(ctx, bt); return rc_; } } while (0)


// J0 / r0 of the windows whose prior lives in a pool slot: device-to-device into the staging k_prior_pack reads
__global__ void __launch_bounds__(256) k_prior_gather(int W, const WinMeta *win, const int *slot, const double *pJ, const double *pr, double *J0s, double *r0s) {
  const int w = blockIdx.x;
  if (w >= W || slot[w] < 0) return;
  const int n = win[w].prior_n;
  const double *sj = pJ + (size_t)slot[w] * 96 * 96, *sr = pr + (size_t)slot[w] * 96;
  for (int e = threadIdx.x; e < n * n; e += 256) J0s[(size_t)w * 96 * 96 + e] = sj[e];
  for (int e = threadIdx.x; e < n; e += 256) r0s[(size_t)w * 96 + e] = sr[e];
}

// MarginalizationFactor (marginalization_factor.cpp:335-395) in normal-equation form, per window: H = J0^T J0 (n x n, ld n),
// b0 = J0^T r0, c0 = r0^T r0, and H scattered into the solver's pre-assembled camera image (PD_* layout).
__global__ void __launch_bounds__(256) k_prior_pack(int W, const WinMeta *win, const double *J0s /*[W][96*96], n x n packed*/,
                                                    const double *r0s /*[W][96]*/, const int *pmap /*[W][96]*/, double *H /*[W][96*96]*/,
                                                    double *b0 /*[W][96]*/, double *c0 /*[W]*/, double *pdense /*[W][PD_N], zeroed*/) {
  extern __shared__ double Jl[];   // J0 of the window (n x n) + r0 (n): read once from HBM, every product out of LDS
  const int w = blockIdx.x, tid = threadIdx.x;
  const int n = win[w].prior_n;
  if (n <= 0) return;
  const double *J = J0s + (size_t)w * 96 * 96, *r = r0s + (size_t)w * 96;
  double *rl = Jl + 96 * 96;
  for (int e = tid; e < n * n; e += 256) Jl[e] = J[e];
  for (int e = tid; e < n; e += 256) rl[e] = r[e];
  __syncthreads();
  const int *pm = pmap + (size_t)w * 96;
  double *Hw = H + (size_t)w * 96 * 96, *pd = pdense + (size_t)w * PD_N;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    if (j > i) continue;
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += Jl[k * n + i] * Jl[k * n + j];
    Hw[(size_t)i * n + j] = s;
    Hw[(size_t)j * n + i] = s;
    for (int rep = 0; rep < (i == j ? 1 : 2); ++rep) {
      const int ci = rep ? pm[j] : pm[i], cq = rep ? pm[i] : pm[j];
      if (ci < CD_B0 && cq < CD_B0) pd[PD_C + ci * PD_CLD + cq] = s;
      else if (ci >= CD_B0 && cq >= CD_B0) pd[PD_AD + ((ci - CD_B0) / 13) * 169 + ((ci - CD_B0) % 13) * 13 + (cq - CD_B0) % 13] = s;
      else if (ci >= CD_B0 && cq < CD_B0) pd[PD_BP + ((ci - CD_B0) % 13) * 80 + cq] = s;
    }
  }
  for (int i = tid; i < n; i += 256) {
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += Jl[k * n + i] * rl[k];
    b0[(size_t)w * 96 + i] = s;
  }
  if (tid == 0) {
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += rl[k] * rl[k];
    c0[w] = s;
  }
}

// camera dim of the first local dim of a prior block id; -1 if unsupported
int prior_block_cd(int id, int *state_off) {
  const int kind = id / 16, index = id % 16;
  switch (kind) {
    case VILO_BLK_POSE: if (index > 10) return -1; *state_off = XO_POSE + 7 * index; return 6 * index;
    case VILO_BLK_SB: if (index > 10) return -1; *state_off = XO_SB + 9 * index; return CD_B0 + 13 * index;
    case VILO_BLK_LB: if (index > 10) return -1; *state_off = XO_LB + 4 * index; return CD_B0 + 13 * index + 9;
    case VILO_BLK_EX: if (index > 1) return -1; *state_off = XO_EX + 7 * index; return CD_EX0 + 6 * index;
    case VILO_BLK_TD: *state_off = XO_TD; return CD_TD;
    default: return -1;
  }
}

}  // namespace

BatchDev *vilo_batch_dev(vilo_batch *bt) { return &bt->d; }
// scratch that lives as long as the batch, out of its arena (other translation units: the marginalisation's per-call buffers — a
// hipMalloc / hipFree pair per buffer costs more than the kernels of a one-window call)
int vilo_batch_scratch(vilo_ctx *ctx, vilo_batch *bt, void **p, size_t bytes) { return dev_alloc_bytes(ctx, bt, p, bytes); }
const int *vilo_batch_perm(vilo_batch *bt, int win, int *L) {
  *L = bt->L_host[win];
  return bt->perm_host.data() + bt->lm_off_host[win];
}

extern "C" void vilo_batch_destroy(vilo_ctx *ctx, vilo_batch *bt) {
  if (!bt) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (bt->gexec) (void)hipGraphExecDestroy(bt->gexec);
  if (ctx) {
    (void)hipStreamSynchronize(ctx->stream);   // nothing of this batch may still be running when its memory is handed on
    for (auto &c : bt->chunks_dev) ctx->pool_free.push_back(c);
  } else {
    for (auto &c : bt->chunks_dev) (void)hipFree(c.first);
  }
  delete bt;
}

int vilo_launch_preint_gather(vilo_ctx *ctx, const vilo_preint_streams *pool, int n, const int *d_ids, const int *d_dst, void *d_out);
int vilo_solver_form(const vilo_ctx *ctx, const BatchDev &b);   // kernels_wave.hip
int vilo_assemble_small_max();                                  // kernels_asm_small.hip
// up to this many packed waves a batch takes the frame-parallel form of the visual linearisation (tuning aids: VILO_TPAR_MAX_WAVES; VILO_NO_TPAR=1 = 0)
static size_t vilo_tpar_max_waves() {
  static const size_t v = [] { const char *e = getenv("VILO_TPAR_MAX_WAVES"); return getenv("VILO_NO_TPAR") ? (size_t)0 : (e ? (size_t)atol(e) : (size_t)256); }();
  return v;
}

extern "C" int vilo_batch_create(vilo_ctx *ctx, int W, const vilo_window_desc *in, const vilo_window_state *init, vilo_batch **out) {
  return vilo_batch_create_refs(ctx, W, in, nullptr, init, out);
}

// refs (optional, [W]): device-resident preintegration objects / prior slots instead of the host records of the descs
int vilo_batch_create_refs(vilo_ctx *ctx, int W, const vilo_window_desc *in, const vilo_resident_refs *refs, const vilo_window_state *init, vilo_batch **out) {
  if (!ctx || !in || !init || !out || W <= 0) return VILO_ERR_BAD_ARG;
  *out = nullptr;
  VILO_HIP(hipSetDevice(ctx->device));
  vilo_batch *bt = new vilo_batch();
  bt->W = W;
  memset(&bt->d, 0, sizeof(BatchDev));
  const bool timing = getenv("VILO_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  double t_prior = 0.0;
  size_t rec_bytes_up = 0;   // bytes of preintegration records that crossed the bus when not the full records'
  std::vector<WinMeta> wins(W);
  std::vector<ChunkMeta> chunks;
  std::vector<WaveMeta> waves;
  std::vector<double> x0((size_t)W * XSTRIDE, 0.0), lam0;
  std::vector<unsigned char> lm_s;   // start frame per landmark (device order)
  std::vector<double> px0((size_t)W * 280, 0.0);
  // J0 / r0 of the priors, packed n x n per window, staged for k_prior_pack (uninitialised storage: only n x n of a slot is read)
  double *pJ = (double *)vilo_host_stage(ctx, 0, sizeof(double) * (size_t)W * 96 * 96), *pr0 = (double *)vilo_host_stage(ctx, 1, sizeof(double) * (size_t)W * 96);
  bool any_prior = false;
  std::vector<unsigned char> iskip((size_t)W * 10, 0);
  std::vector<int> pmap((size_t)W * 96, 0), pbs((size_t)W * 40, 0), pbi((size_t)W * 40, 0), pbx((size_t)W * 40, 0), pbst((size_t)W * 40, 0);
  int lm_total = 0, gram_total = 0;
  size_t obs_total = 0, flags_total = 0;
  bt->lm_off_host.resize(W);
  bt->L_host.resize(W);

  // ---- pass 1 (serial, light): validation, chunk / wave tables and every offset ----
  for (int w = 0; w < W; ++w) {
    const vilo_window_desc &d = in[w];
    const vilo_window_state &s = init[w];
    if (d.n_frames < 2 || d.n_frames > VILO_MAX_FRAMES || d.n_landmarks < 0 || d.n_landmarks > VILO_NUM_OF_F) {
      ctx->err = "window sizes out of range"; vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG;
    }
    if ((d.use_leg != 0) != (in[0].use_leg != 0)) { ctx->err = "all windows of a batch must use the same IMU factor kind (use_leg)"; vilo_batch_destroy(ctx, bt); return VILO_ERR_UNSUPPORTED; }
    const vilo_resident_refs *rf = refs ? refs + w : nullptr;
    if (rf && rf->preint_pool && ((d.use_leg != 0) == (rf->preint_pool->kind != 0) || !rf->preint_ids || !rf->preint_sum_dt || rf->preint_pool != refs[0].preint_pool)) {
      ctx->err = "resident preintegration: pool kind must match use_leg, one pool per batch"; vilo_batch_destroy(ctx, bt); return VILO_ERR_UNSUPPORTED;
    }
    if (rf && rf->prior_pool && rf->prior_pool != refs[0].prior_pool) { ctx->err = "one prior pool per batch"; vilo_batch_destroy(ctx, bt); return VILO_ERR_UNSUPPORTED; }
    if (!d.use_leg && !d.preint_imu && !(rf && rf->preint_pool)) { vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG; }
    if ((d.use_leg && !d.preint && !(rf && rf->preint_pool)) || !s.pose || !s.speed_bias || !s.leg_bias || !s.ex_pose || !s.td || (d.n_landmarks && (!s.inv_depth || !d.lm_start_frame || !d.lm_obs_offset || !d.obs || !d.obs_is_stereo))) {
      vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG;
    }
    const int F = d.n_frames, L = d.n_landmarks;
    // the observation table is indexed through lm_obs_offset: [0] = 0, non-decreasing, [L] = n_obs — a window that comes out of a file
    // (vilo_window_io.h) is untrusted, and the packing below reads obs[11 * (offset + t)] for t < K
    if (L > 0) {
      bool ok = d.n_obs >= 0 && d.lm_obs_offset[0] == 0 && d.lm_obs_offset[L] == d.n_obs;
      for (int l = 0; ok && l < L; ++l) ok = d.lm_obs_offset[l + 1] >= d.lm_obs_offset[l] && d.lm_start_frame[l] >= 0 && d.lm_start_frame[l] < F;
      if (!ok) { ctx->err = "landmark observation table: lm_obs_offset must start at 0, not decrease and end at n_obs; start frames must lie in the window"; vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG; }
    }
    WinMeta &wm = wins[w];
    memset(&wm, 0, sizeof(wm));
    wm.n_frames = F; wm.L = L; wm.use_leg = d.use_leg; wm.pad = -1;
    wm.lm_off = lm_total; wm.chunk_off = (int)chunks.size(); wm.gram_off = gram_total;
    // use_leg == 0: the leg-bias blocks are not part of the problem (estimator.cpp:1071-1072): masked like constant blocks
    wm.const_mask = ((d.leg_bias_const || !d.use_leg) ? CONST_LB : 0) | (d.ex_const ? CONST_EX : 0) | (d.td_const ? CONST_TD : 0);
    bt->lm_off_host[w] = lm_total;
    bt->L_host[w] = L;
    // landmark chunks: group by start frame (a stable counting sort: list order preserved inside a group), <= 64 per chunk. The device
    // order of the window's landmarks is perm_host[lm_total ..): a chunk's landmarks are its lm_off .. lm_off + n entries of it.
    int local = 0;
    {
      int first[VILO_MAX_FRAMES + 1] = {0};
      for (int l = 0; l < L; ++l) {
        const int sf = d.lm_start_frame[l];
        if (sf < 0 || sf >= F) { ctx->err = "landmark start_frame outside the window"; vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG; }
        ++first[sf + 1];
      }
      for (int sf = 0; sf < F; ++sf) first[sf + 1] += first[sf];
      const size_t base = bt->perm_host.size();
      bt->perm_host.resize(base + L); lam0.resize(base + L); lm_s.resize(base + L);
      int fill[VILO_MAX_FRAMES];
      for (int sf = 0; sf < F; ++sf) fill[sf] = first[sf];
      for (int l = 0; l < L; ++l) {
        const int sf = d.lm_start_frame[l], at = fill[sf]++;
        bt->perm_host[base + at] = l; lam0[base + at] = s.inv_depth[l]; lm_s[base + at] = (unsigned char)sf;
      }
      for (int sf = 0; sf < F; ++sf) {
        for (int c0 = first[sf]; c0 < first[sf + 1]; c0 += 64) {
          const int n = std::min(64, first[sf + 1] - c0);
          ChunkMeta cm;
          memset(&cm, 0, sizeof(cm));
          cm.win = w; cm.s = sf; cm.n = n; cm.lm_off = lm_total + local; cm.lm_local = local;
          int kmax = 0;
          for (int i = 0; i < n; ++i) {
            const int l = bt->perm_host[base + c0 + i];
            const int K = d.lm_obs_offset[l + 1] - d.lm_obs_offset[l];
            if (K < 1 || sf + K > F) { ctx->err = "landmark observation range outside the window"; vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG; }
            kmax = std::max(kmax, K);
          }
          cm.kmax = kmax;
          cm.gram_off = gram_total;
          gram_total += kmax;
          local += n;
          chunks.push_back(cm);
        }
      }
    }
    if (local != L) { ctx->err = "landmark start_frame outside the window"; vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG; }
    wm.n_chunks = (int)chunks.size() - wm.chunk_off;
    // pack the window's chunks into waves: consecutive chunks side by side, each at a lane multiple of 8, <= 4 per wave
    wm.wave_off = (int)waves.size();
    for (int c = wm.chunk_off; c < (int)chunks.size();) {
      WaveMeta wv;
      memset(&wv, 0, sizeof(wv));
      wv.win = w;
      int lanes = 0;
      while (c < (int)chunks.size() && wv.nseg < 4) {
        const int pad = (chunks[c].n + 7) & ~7;
        if (lanes + pad > 64) break;
        wv.seg_chunk[wv.nseg] = c; wv.seg_lane0[wv.nseg] = lanes;
        wv.kmax = std::max(wv.kmax, chunks[c].kmax);
        lanes += pad; ++wv.nseg; ++c;
      }
      wv.n_lanes = lanes;
      wv.obs_off = (long long)obs_total;
      wv.flag_off = (long long)flags_total;
      obs_total += (size_t)wv.kmax * 11 * lanes;
      flags_total += (size_t)wv.kmax * lanes;
      waves.push_back(wv);
    }
    wm.n_waves = (int)waves.size() - wm.wave_off;
    wm.n_gram = gram_total - wm.gram_off;
    lm_total += L;
    const vilo_prior *pr = vilo_win_prior(d, rf);
    if (pr && pr->valid && pr->n > 0) {
      if (pr->n > VILO_MAX_PRIOR_DIM || pr->n_blocks < 0 || pr->n_blocks > VILO_MAX_PRIOR_BLOCKS) { vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG; }
      // block tables: sizes are the global sizes the estimator uses (1, 4, 7, 9), a block's local rows lie inside the prior, its id
      // names a block kind and index that exist (kind * 16 + index; the assembly indexes its LDS image through these)
      for (int k = 0; k < pr->n_blocks; ++k) {
        const int gs = pr->block_size[k], ls = gs == 7 ? 6 : gs, idx = pr->block_idx[k], id = pr->block_id[k];
        static const int kind_size[5] = {7, 9, 4, 7, 1};   // VILO_BLK_POSE, _SB, _LB, _EX, _TD
        if (id < 0 || id >= 16 * 5) continue;   // (not a camera-side block — e.g. a feature: refused as unsupported, with its own message, by the packing pass)
        if (gs != kind_size[id >> 4] || (id & 15) >= ((id >> 4) == VILO_BLK_EX ? 2 : (id >> 4) == VILO_BLK_TD ? 1 : VILO_MAX_FRAMES) || idx < 0 || idx + ls > pr->n) {
          ctx->err = "prior block table out of range"; vilo_batch_destroy(ctx, bt); return VILO_ERR_BAD_ARG;
        }
      }
      any_prior = true;
    }
  }
  t_prior = now() - t_begin;   // (pass 1's share of the packing time, for VILO_HOST_TIMING)
  // ---- pass 2 (one host thread per slice of windows): the heavy copies — wave-packed observation image, states, prior staging ----
  double *obs = (double *)vilo_host_stage(ctx, 2, sizeof(double) * std::max<size_t>(1, obs_total));
  unsigned char *flags = (unsigned char *)vilo_host_stage(ctx, 3, std::max<size_t>(1, flags_total));
  if (!pJ || !pr0 || !obs || !flags) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
  std::vector<int> win_err(W, 0);
  auto fill_window = [&](int w) {
    const vilo_window_desc &d = in[w];
    const vilo_window_state &s = init[w];
    WinMeta &wm = wins[w];
    const int F = d.n_frames;
    // states (vector2double layout)
    double *xw = &x0[(size_t)w * XSTRIDE];
    memcpy(xw + XO_POSE, s.pose, sizeof(double) * 7 * F);
    memcpy(xw + XO_SB, s.speed_bias, sizeof(double) * 9 * F);
    memcpy(xw + XO_LB, s.leg_bias, sizeof(double) * 4 * F);
    for (int k = F; k < VILO_MAX_FRAMES; ++k) xw[XO_POSE + 7 * k + 6] = 1.0;
    memcpy(xw + XO_EX, s.ex_pose, sizeof(double) * 14);
    xw[XO_TD] = s.td[0];
    for (int wi = wm.wave_off; wi < wm.wave_off + wm.n_waves; ++wi) {
      const WaveMeta &wv = waves[wi];
      const int lanes = wv.n_lanes;
      double *ob = obs + wv.obs_off;
      unsigned char *fl = flags + wv.flag_off;
      memset(ob, 0, sizeof(double) * (size_t)wv.kmax * 11 * lanes);
      memset(fl, 0, (size_t)wv.kmax * lanes);
      for (int g = 0; g < wv.nseg; ++g) {
        const ChunkMeta &cm = chunks[wv.seg_chunk[g]];
        const int *ids = bt->perm_host.data() + cm.lm_off;   // the chunk's landmarks (window order)
        for (int i = 0; i < cm.n; ++i) {
          const int l = ids[i], lane = wv.seg_lane0[g] + i;
          const int o0 = d.lm_obs_offset[l], K = d.lm_obs_offset[l + 1] - o0;
          for (int t = 0; t < K; ++t) {
            for (int f = 0; f < 11; ++f) ob[((size_t)t * 11 + f) * lanes + lane] = d.obs[(size_t)(o0 + t) * 11 + f];
            fl[(size_t)t * lanes + lane] = (unsigned char)(1 | (d.obs_is_stereo[o0 + t] ? 2 : 0));
          }
        }
      }
    }
    const vilo_resident_refs *rf = refs ? refs + w : nullptr;
    for (int k = 0; k < 10; ++k) iskip[(size_t)w * 10 + k] = (k + 1 < F && !(vilo_win_sum_dt(d, rf, k) > 10.0)) ? 0 : 1;
    // prior (MarginalizationFactor, marginalization_factor.cpp:335-395): block tables here, H = J0^T J0, b0 = J0^T r0, c0 = r0^T r0 in k_prior_pack
    const vilo_prior *pr = vilo_win_prior(d, rf);
    if (pr && pr->valid && pr->n > 0) {
      const vilo_prior &p = *pr;
      const int n = p.n;
      wm.prior_n = n; wm.prior_nb = p.n_blocks;
      int xo = 0, bframe = -1;
      for (int k = 0; k < p.n_blocks; ++k) {
        int soff = 0;
        const int cd = prior_block_cd(p.block_id[k], &soff);
        const int gs = p.block_size[k], ls = gs == 7 ? 6 : gs;
        static const int kind_size[5] = {7, 9, 4, 7, 1};   // global sizes of VILO_BLK_POSE / SB / LB / EX / TD
        const int kind = p.block_id[k] / 16;
        if (p.block_id[k] < 0 || kind < 0 || kind > 4 || gs != kind_size[kind] || xo + gs > 280) { win_err[w] = 1; return; }
        if (cd < 0 || p.block_idx[k] < 0 || p.block_idx[k] + ls > n) { win_err[w] = 1; return; }
        for (int q = 0; q < k; ++q) {   // local index ranges must not overlap
          const int lq = p.block_size[q] == 7 ? 6 : p.block_size[q];
          if (p.block_idx[k] < p.block_idx[q] + lq && p.block_idx[q] < p.block_idx[k] + ls) { win_err[w] = 1; return; }
        }
        if (cd >= CD_B0) {
          const int fr = (cd - CD_B0) / 13;
          if (bframe >= 0 && bframe != fr) { win_err[w] = 2; return; }
          bframe = fr;
        }
        pbs[(size_t)w * 40 + k] = gs; pbi[(size_t)w * 40 + k] = p.block_idx[k]; pbx[(size_t)w * 40 + k] = xo; pbst[(size_t)w * 40 + k] = soff;
        for (int c = 0; c < ls; ++c) pmap[(size_t)w * 96 + p.block_idx[k] + c] = cd + c;
        for (int c = 0; c < gs; ++c) px0[(size_t)w * 280 + xo + c] = p.x0[xo + c];
        xo += gs;
      }
      wm.pad = bframe;
      if (p.J0) {   // a pool slot keeps J0 / r0 on the device: gathered there, below
        memcpy(pJ + (size_t)w * 96 * 96, p.J0, sizeof(double) * (size_t)n * n);
        memcpy(pr0 + (size_t)w * 96, p.r0, sizeof(double) * n);
      }
    }
  };
  vilo::parallel_items(W, 8, fill_window, ctx->pool);   // (worker_pool.hpp: host threads parked between batches)
  for (int w = 0; w < W; ++w)
    if (win_err[w]) {
      ctx->err = win_err[w] == 1 ? "unsupported prior block" : "prior couples speed/leg biases of two frames";
      vilo_batch_destroy(ctx, bt);
      return VILO_ERR_UNSUPPORTED;
    }
  const double t_packed = now();
  // a lane of vilo_solve_windows' pipeline: the uploads of the lanes go one after the other at the link's rate (side by side every lane's
  // solve would start when ALL uploads are through); held until this batch's uploads are complete, i.e. to the end of the call
  std::unique_lock<std::mutex> dma_turn;
  if (ctx->dma_turn) dma_turn = std::unique_lock<std::mutex>(*ctx->dma_turn);
  BatchDev &D = bt->d;
  D.W = W; D.n_chunks = (int)chunks.size(); D.n_lm = lm_total; D.n_gram = gram_total; D.n_waves = (int)waves.size();
  // compact visual rows / Gram slots in the solve passes: td must be a constant block in every window (estimate_td: 0, all of the
  // reference's configurations); vilo_set_compact_rows(ctx, 0) keeps the 23-column form
  D.compact = ctx->compact_rows;
  for (int w = 0; w < W; ++w)
    if (!(wins[w].const_mask & CONST_TD)) D.compact = 0;
  UploadBlob blob;
  TRYB(blob.add(ctx, bt, &D.win, wins));
  TRYB(blob.add(ctx, bt, &D.chunk, chunks));
  TRYB(blob.add(ctx, bt, &D.wave, waves));
  {
    // launch order of the packed waves: by decreasing number of frames walked. A single-wave workgroup can only start on the SIMD the
    // dispatcher's cyclic pointer names, so waves of mixed length in flight on one CU leave SIMDs idle behind a long one (measured: 2.7
    // instead of 4 resident waves per CU); with equal lengths adjacent they retire in launch order and the longest ones do not form the tail.
    std::vector<int> order(waves.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    const char *wo_env = getenv("VILO_WAVE_ORDER");   // tuning aid: 0 = window order, 1 = by length, 2 = by length, groups rotated
    const int wo = wo_env ? atoi(wo_env) : 1;
    if (wo >= 1) std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return waves[a].kmax > waves[c].kmax; });
    if (wo == 2) {
      size_t g0 = 0; int gi = 0;
      while (g0 < order.size()) {
        size_t g1 = g0;
        while (g1 < order.size() && waves[order[g1]].kmax == waves[order[g0]].kmax) ++g1;
        if (g1 - g0 > 8) std::rotate(order.begin() + g0, order.begin() + g0 + (gi % 8), order.begin() + g1);
        g0 = g1; ++gi;
      }
    }
    TRYB(blob.add(ctx, bt, &D.wave_order, order));
  }
  TRYB(blob.add(ctx, bt, &D.obs, obs, obs_total));
  TRYB(blob.add(ctx, bt, &D.flags, flags, flags_total));
  TRYB(blob.add(ctx, bt, &D.x0, x0));
  TRYB(blob.add(ctx, bt, &D.lam0, lam0));
  TRYB(blob.add(ctx, bt, &D.lm_s, lm_s));
  TRYB(blob.add(ctx, bt, &D.lm_perm, bt->perm_host));
  TRYB(dev_alloc(ctx, bt, &D.x, (size_t)W * XSTRIDE));
  TRYB(dev_alloc(ctx, bt, &D.xc, (size_t)W * XSTRIDE));
  TRYB(dev_alloc(ctx, bt, &D.lam, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lamc, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_E, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_gbuf[0], (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_gbuf[1], (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_dh2, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_y, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_scale, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_einv, (size_t)lm_total));
  TRYB(dev_alloc(ctx, bt, &D.lm_w, (size_t)lm_total * 80));
  // a landmark's coupling rows with the poses before its start frame are structural zeros: written here once, never again
  if (hipMemsetAsync(D.lm_w, 0, sizeof(double) * (size_t)std::max(lm_total, 1) * 80, ctx->stream) != hipSuccess) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
  // few windows: one workgroup per (packed wave, frame) instead of per packed wave, so that the chip is not left to 3 waves per window
  // (the landmark-side terms every such workgroup writes: all of them, zeros included — nothing reads an entry nobody wrote)
  D.lm_part = nullptr;
  const size_t tpar_max = vilo_tpar_max_waves();
  D.full_regime = ctx->regime_full;
  if (waves.size() <= tpar_max && !ctx->regime_full) TRYB(dev_alloc(ctx, bt, &D.lm_part, (size_t)lm_total * VILO_MAX_FRAMES * 2 * 21));
  TRYB(dev_alloc(ctx, bt, &D.gram, (size_t)gram_total * VILO_GRAM));
  TRYB(dev_alloc(ctx, bt, &D.chunk_cost, waves.size() * VILO_MAX_FRAMES));   // per (packed wave, frame offset) partial costs
  TRYB(dev_alloc(ctx, bt, &D.prep, (size_t)W * 10));
  TRYB(dev_alloc(ctx, bt, &D.imu_lin, (size_t)W * 10 * 31 * 39));
  TRYB(dev_alloc(ctx, bt, &D.imu_raw, (size_t)W * 10 * 31 * 39));
  if (hipMemsetAsync(D.imu_raw, 0, sizeof(double) * (size_t)W * 10 * 31 * 39, ctx->stream) != hipSuccess) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
  TRYB(dev_alloc(ctx, bt, &D.imu_gram, (size_t)W * 10 * 780));
  TRYB(dev_alloc(ctx, bt, &D.imu_cost, (size_t)W * 10));
  TRYB(blob.add(ctx, bt, &D.imu_skip, iskip));
  TRYB(dev_alloc(ctx, bt, &D.prior_H, (size_t)W * 96 * 96));
  TRYB(dev_alloc(ctx, bt, &D.prior_dense, (size_t)W * PD_N));
  TRYB(dev_alloc(ctx, bt, &D.prior_hd, (size_t)W * 96));
  TRYB(dev_alloc(ctx, bt, &D.prior_b0, (size_t)W * 96));
  TRYB(dev_alloc(ctx, bt, &D.prior_c0, (size_t)W));
  TRYB(blob.add(ctx, bt, &D.prior_x0, px0));
  TRYB(blob.add(ctx, bt, &D.prior_map, pmap));
  if (hipMemsetAsync(D.prior_dense, 0, sizeof(double) * (size_t)W * PD_N, ctx->stream) != hipSuccess ||
      hipMemsetAsync(D.prior_b0, 0, sizeof(double) * (size_t)W * 96, ctx->stream) != hipSuccess ||
      hipMemsetAsync(D.prior_c0, 0, sizeof(double) * (size_t)W, ctx->stream) != hipSuccess ||
      hipMemsetAsync(D.prior_H, 0, sizeof(double) * (size_t)W * 96 * 96, ctx->stream) != hipSuccess) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
  TRYB(blob.add(ctx, bt, &D.prior_bsize, pbs));
  TRYB(blob.add(ctx, bt, &D.prior_bidx, pbi));
  TRYB(blob.add(ctx, bt, &D.prior_bxoff, pbx));
  TRYB(blob.add(ctx, bt, &D.prior_bstate, pbst));
  TRYB(blob.flush(ctx, bt));   // D.win ... D.prior_bstate are device pointers from here on
  TRYB(dev_alloc(ctx, bt, &D.cam_g, (size_t)W * CD_N));
  TRYB(dev_alloc(ctx, bt, &D.cam_dh2, (size_t)W * CD_N));
  TRYB(dev_alloc(ctx, bt, &D.cam_y, (size_t)W * CD_N));
  TRYB(dev_alloc(ctx, bt, &D.cam_scale, (size_t)W * CD_N));
  TRYB(dev_alloc(ctx, bt, &D.Lk, (size_t)W * 11 * 169));
  TRYB(dev_alloc(ctx, bt, &D.TAg, (size_t)W * 11 * 169));
  TRYB(dev_alloc(ctx, bt, &D.Cimg, (size_t)W * 3840));
  TRYB(dev_alloc(ctx, bt, &D.Tk, (size_t)W * TK_N));
  TRYB(dev_alloc(ctx, bt, &D.cam_gin, (size_t)W * CD_N));
  TRYB(dev_alloc(ctx, bt, &D.Bimg, (size_t)W * BI_N));
  TRYB(dev_alloc(ctx, bt, &D.st, (size_t)W));
  TRYB(dev_alloc(ctx, bt, &D.status, 1));
  TRYB(dev_alloc(ctx, bt, &D.lin_cur, (size_t)W));
  if (hipMemsetAsync(D.status, 0, sizeof(int), ctx->stream) != hipSuccess || hipMemsetAsync(D.st, 0, sizeof(SolverState) * (size_t)W, ctx->stream) != hipSuccess) {
    vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP;
  }
  if (any_prior) {
    double *d_J = nullptr, *d_r = nullptr;
    TRYB(dev_alloc(ctx, bt, &d_J, (size_t)W * 96 * 96));
    TRYB(dev_alloc(ctx, bt, &d_r, (size_t)W * 96));
    if (hipMemcpyAsync(d_J, pJ, sizeof(double) * (size_t)W * 96 * 96, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_r, pr0, sizeof(double) * (size_t)W * 96, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
    if (refs && refs[0].prior_pool) {
      std::vector<int> slot(W, -1);
      for (int w = 0; w < W; ++w)
        if (refs[w].prior_pool && refs[w].prior_slot >= 0 && refs[w].prior_slot < refs[w].prior_pool->n && wins[w].prior_n > 0) slot[w] = refs[w].prior_slot;
      int *d_slot = nullptr;
      TRYB(dev_upload(ctx, bt, &d_slot, slot));
      hipLaunchKernelGGL(k_prior_gather, dim3(W), dim3(256), 0, ctx->stream, W, D.win, d_slot, refs[0].prior_pool->dJ, refs[0].prior_pool->dr, d_J, d_r);
    }
    {
      const size_t pp_lds = sizeof(double) * (96 * 96 + 96);
      if (!ctx->prior_attr_set) {
        if (hipFuncSetAttribute((const void *)k_prior_pack, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp_lds) != hipSuccess) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
        ctx->prior_attr_set = true;
      }
      hipLaunchKernelGGL(k_prior_pack, dim3(W), dim3(256), pp_lds, ctx->stream, W, D.win, d_J, d_r, D.prior_map, D.prior_H, D.prior_b0, D.prior_c0, D.prior_dense);
    }
    if (hipGetLastError() != hipSuccess) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }
  }
  const double t_uploaded = now();
  // hoist sqrt_info = chol(cov^-1)^T out of the iteration loop (the reference recomputes it on every
  // IMULegFactor::Evaluate, imu_leg_factor.cpp:197-198)
  {
    const bool leg = in[0].use_leg != 0;
    void *d_pre = nullptr;
    const size_t rec = leg ? sizeof(vilo_preint) : sizeof(vilo_preint_imu);
    const size_t bytes = rec * (size_t)W * 10;
    if (dev_alloc_bytes(ctx, bt, &d_pre, bytes) != VILO_OK) { vilo_batch_destroy(ctx, bt); return VILO_ERR_HIP; }   // arena: lives as long as the batch
    int rc = VILO_OK;
    // records go from the caller's arrays straight to the device (a staged host copy of W x 156 KB costs more than the W copies)
    bool partial = false;
    for (int w = 0; w < W; ++w) partial = partial || in[w].n_frames < VILO_MAX_FRAMES;
    if (partial && hipMemsetAsync(d_pre, 0, bytes, ctx->stream) != hipSuccess) rc = VILO_ERR_HIP;
    std::vector<int> g_ids, g_dst, host_rec;   // host_rec: windows whose records come out of the caller's arrays
    for (int w = 0; w < W && rc == VILO_OK; ++w) {
      if (refs && refs[w].preint_pool) {
        for (int k = 0; k + 1 < in[w].n_frames; ++k) {
          const int id = refs[w].preint_ids[k];
          if (id < 0 || id >= refs[w].preint_pool->n) { rc = VILO_ERR_BAD_ARG; break; }
          g_ids.push_back(id); g_dst.push_back(w * 10 + k);
        }
        continue;
      }
      host_rec.push_back(w);
    }
    static const bool no_compact_rec = getenv("VILO_FULL_RECORD_UPLOAD") != nullptr;
    if (rc == VILO_OK && leg && (int)host_rec.size() == W && ctx->sqrt_info_mode == 0 && !no_compact_rec) {
      // every window's records from host memory, default sqrt_info route: the compact form (above) through the two page-locked chunks, the
      // worker pool gathering chunk k + 1 while the DMA engine reads chunk k; the device expands each chunk behind its copy
      const size_t per_win = sizeof(double) * REC_C_N * 10, chunk_w = std::max<size_t>(1, ((size_t)32 << 20) / per_win);
      char *ring[2] = {(char *)vilo_host_stage(ctx, 4, per_win * chunk_w), (char *)vilo_host_stage(ctx, 5, per_win * chunk_w)};
      if (!ctx->rec_ev[0]) { (void)hipEventCreateWithFlags(&ctx->rec_ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ctx->rec_ev[1], hipEventDisableTiming); }
      void *d_comp = nullptr;
      if (!ring[0] || !ring[1] || !ctx->rec_ev[0] || !ctx->rec_ev[1] || dev_alloc_bytes(ctx, bt, &d_comp, per_win * (size_t)W) != VILO_OK) rc = VILO_ERR_HIP;
      bool used[2] = {false, false};
      for (size_t c0 = 0, ci = 0; rc == VILO_OK && c0 < (size_t)W; c0 += chunk_w, ++ci) {
        const int sl = (int)(ci & 1);
        const size_t cn = std::min(chunk_w, (size_t)W - c0);
        if (used[sl] && hipEventSynchronize(ctx->rec_ev[sl]) != hipSuccess) { rc = VILO_ERR_HIP; break; }
        vilo::parallel_items((int)cn, 4, [&](int i) {
          const int w = (int)c0 + i, nr = in[w].n_frames - 1;
          double *dst = (double *)(ring[sl] + per_win * (size_t)i);
          for (int k = 0; k < nr; ++k) rec_compact(in[w].preint + k, dst + (size_t)REC_C_N * k);
          if (nr < 10) memset(dst + (size_t)REC_C_N * nr, 0, sizeof(double) * REC_C_N * (size_t)(10 - nr));   // (intervals the window does not have: zero records, as before)
        }, ctx->pool);
        if (hipMemcpyAsync((char *)d_comp + per_win * c0, ring[sl], per_win * cn, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = VILO_ERR_HIP; break; }
        if (hipEventRecord(ctx->rec_ev[sl], ctx->stream) != hipSuccess) { rc = VILO_ERR_HIP; break; }
        used[sl] = true;
        hipLaunchKernelGGL(k_expand_records, dim3((unsigned)(cn * 10)), dim3(256), 0, ctx->stream, (int)(cn * 10), (const double *)d_comp + (size_t)REC_C_N * 10 * c0,
                           (vilo_preint *)d_pre + 10 * c0);
      }
      if (rc == VILO_OK && hipGetLastError() != hipSuccess) rc = VILO_ERR_HIP;
      rec_bytes_up = per_win * (size_t)W;
    } else if (rc == VILO_OK && !host_rec.empty()) {
      // The caller's records (156 KB per window, pageable) go up through two page-locked chunks of the context: the worker pool copies
      // chunk k + 1 into one while the DMA engine reads chunk k out of the other. (One hipMemcpyAsync per window straight from the caller's
      // arrays moved 640 MB of a 4096-window batch at 11.5 GB/s — the runtime's own bounce buffer, one thread.)
      const size_t per_win = rec * 10, chunk_w = std::max<size_t>(1, ((size_t)32 << 20) / per_win);
      char *ring[2] = {(char *)vilo_host_stage(ctx, 4, per_win * chunk_w), (char *)vilo_host_stage(ctx, 5, per_win * chunk_w)};
      if (!ctx->rec_ev[0]) { (void)hipEventCreateWithFlags(&ctx->rec_ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ctx->rec_ev[1], hipEventDisableTiming); }
      bool used[2] = {false, false};
      if (!ring[0] || !ring[1] || !ctx->rec_ev[0] || !ctx->rec_ev[1]) rc = VILO_ERR_HIP;
      for (size_t c0 = 0, ci = 0; rc == VILO_OK && c0 < host_rec.size(); c0 += chunk_w, ++ci) {
        const int sl = (int)(ci & 1);
        const size_t cn = std::min(chunk_w, host_rec.size() - c0);
        if (used[sl] && hipEventSynchronize(ctx->rec_ev[sl]) != hipSuccess) { rc = VILO_ERR_HIP; break; }
        vilo::parallel_items((int)cn, 4, [&](int i) {
          const int w = host_rec[c0 + i];
          const void *src = leg ? (const void *)in[w].preint : (const void *)in[w].preint_imu;
          memcpy(ring[sl] + per_win * (size_t)i, src, rec * (size_t)(in[w].n_frames - 1));
        }, ctx->pool);
        // consecutive windows of the chunk that are consecutive in the batch go up in one copy
        for (size_t i = 0; i < cn && rc == VILO_OK;) {
          size_t j = i + 1;
          while (j < cn && host_rec[c0 + j] == host_rec[c0 + j - 1] + 1 && in[host_rec[c0 + j - 1]].n_frames == VILO_MAX_FRAMES) ++j;
          const size_t bytes_run = (j - i - 1) * per_win + rec * (size_t)(in[host_rec[c0 + j - 1]].n_frames - 1);
          if (hipMemcpyAsync((char *)d_pre + per_win * (size_t)host_rec[c0 + i], ring[sl] + per_win * i, bytes_run, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = VILO_ERR_HIP;
          i = j;
        }
        if (rc == VILO_OK && hipEventRecord(ctx->rec_ev[sl], ctx->stream) != hipSuccess) rc = VILO_ERR_HIP;
        used[sl] = true;
      }
    }
    if (rc == VILO_OK && !g_ids.empty()) {
      int *d_gi = nullptr, *d_gd = nullptr;
      rc = dev_upload(ctx, bt, &d_gi, g_ids);
      if (rc == VILO_OK) rc = dev_upload(ctx, bt, &d_gd, g_dst);
      if (rc == VILO_OK) rc = vilo_launch_preint_gather(ctx, refs[0].preint_pool, (int)g_ids.size(), d_gi, d_gd, d_pre);
    }
    bt->d_pre = d_pre; bt->leg = leg;
    if (rc == VILO_OK) rc = dev_alloc(ctx, bt, &bt->d_prep_bad, (size_t)W * 10);
    D.prep_bad = bt->d_prep_bad;   // (per-interval flags of the records in force: the marginalisation looks at the intervals it uses)
    if (rc == VILO_OK) rc = vilo_batch_prepare(ctx, bt);
    // a covariance that is not positive definite has no sqrt_info: that window alone fails (termination FAILURE, like a non-finite
    // IterationZero); the flag is looked at for live intervals only — folded per window on the device (no round trip through the host:
    // a one-window batch is built for every image of a replay)
    if (rc == VILO_OK) rc = dev_alloc(ctx, bt, &D.win_bad, (size_t)W);
    if (rc == VILO_OK) {
      hipLaunchKernelGGL(k_fold_win_bad, dim3((W + 255) / 256), dim3(256), 0, ctx->stream, W, bt->d_prep_bad, D.imu_skip, D.win_bad);
      if (hipGetLastError() != hipSuccess) rc = VILO_ERR_HIP;
    }
    // (the uploads above came out of this call's host vectors and the context's reusable staging: they are complete when it returns)
    if (rc == VILO_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = VILO_ERR_HIP;
    if (rc != VILO_OK) { vilo_batch_destroy(ctx, bt); return rc; }
  }
  const double t_prep = now();
  int rc = vilo_batch_reset(ctx, bt);
  if (rc != VILO_OK) { vilo_batch_destroy(ctx, bt); return rc; }
  ctx->last_create_ms[0] = now() - t_begin; ctx->last_create_ms[1] = t_packed - t_begin; ctx->last_create_ms[2] = t_uploaded - t_packed; ctx->last_create_ms[3] = t_prep - t_uploaded;
  ctx->last_create_bytes = (double)(sizeof(double) * (obs_total + (size_t)W * XSTRIDE + (any_prior ? (size_t)W * (96 * 96 + 96) : 0)) + flags_total +
                                    (rec_bytes_up ? rec_bytes_up : (in[0].use_leg ? sizeof(vilo_preint) : sizeof(vilo_preint_imu)) * (size_t)W * 10));
  if (timing)
    fprintf(stderr, "[vilo_batch_create] W=%d pack %.2f ms (of which the serial table pass %.2f) alloc+upload %.2f ms preint %.2f ms reset %.2f ms\n", W, t_packed - t_begin, t_prior,
            t_uploaded - t_packed, t_prep - t_uploaded, now() - t_prep);
  *out = bt;
  return VILO_OK;
}

// sqrt_info = chol(cov^-1)^T of every live interval of the batch (asynchronous, on the context's stream)
extern "C" int vilo_batch_prepare(vilo_ctx *ctx, vilo_batch *bt) {
  if (!ctx || !bt || !bt->d_pre) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  BatchDev &D = bt->d;
  VILO_HIP(hipMemsetAsync(bt->d_prep_bad, 0, sizeof(int) * (size_t)bt->W * 10, ctx->stream));
  const bool timed = ctx->profile == 1 || ctx->profile == 2 + 11;   // (kind 11 = k_prepare_preint, vilo_kernel_name)
  if (timed) {
    for (hipEvent_t &e : ctx->prep_ev) if (!e) VILO_HIP(hipEventCreate(&e));
    VILO_HIP(hipEventRecord(ctx->prep_ev[0], ctx->stream));
  }
  int rc = bt->leg ? vilo_launch_prepare_preint(ctx, bt->W * 10, (const vilo_preint *)bt->d_pre, D.prep, bt->d_prep_bad, D.imu_skip, 1)
                   : vilo_launch_prepare_preint_imu(ctx, bt->W * 10, (const vilo_preint_imu *)bt->d_pre, D.prep, bt->d_prep_bad, D.imu_skip, 1);
  if (rc == VILO_OK && !bt->leg) rc = vilo_launch_embed_sqrt15(ctx, D);
  if (timed) {
    VILO_HIP(hipEventRecord(ctx->prep_ev[1], ctx->stream));
    ctx->prep_pending = true;
  }
  return rc;
}

// BASELINE configs[2]: keep the samples behind the batch's IMU-leg records in HBM and integrate every live interval again
// (IMULegIntegrationBase::repropagate, imu_leg_integration_base.cpp:62-86) at the biases of each point the solver linearises
extern "C" int vilo_batch_set_samples(vilo_ctx *ctx, vilo_batch *bt, const vilo_sample *samples, const int32_t *offsets) {
  if (!ctx || !bt) return VILO_ERR_BAD_ARG;
  BatchDev &D = bt->d;
  if (!samples) {   // back to records integrated once
    if (!D.rp_on) return VILO_OK;
    D.rp_on = 0;
    if (bt->gexec) { (void)hipGraphExecDestroy(bt->gexec); bt->gexec = nullptr; }
    // d_pre holds what the last re-integration left (the records at the last candidate point): the records the batch was created with come
    // back from their copy, their sqrt_info is prepared again and win_bad is rebuilt from the flags of that preparation — a covariance
    // without sqrt_info fails its window again, as it did before the samples were set
    VILO_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)bt->W * 10;
    if (bt->rp_orig) VILO_HIP(hipMemcpyAsync(bt->d_pre, bt->rp_orig, sizeof(vilo_preint) * n, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = vilo_batch_prepare(ctx, bt);
    if (rc != VILO_OK) return rc;
    std::vector<int> bad(n, 0), winbad(bt->W, 0);
    std::vector<unsigned char> skip(n);
    VILO_HIP(hipStreamSynchronize(ctx->stream));
    VILO_HIP(hipMemcpy(bad.data(), bt->d_prep_bad, sizeof(int) * n, hipMemcpyDeviceToHost));
    VILO_HIP(hipMemcpy(skip.data(), D.imu_skip, n, hipMemcpyDeviceToHost));
    for (size_t f = 0; f < n; ++f)
      if (!skip[f] && bad[f]) winbad[f / 10] = 1;
    if (D.win_bad) VILO_HIP(hipMemcpy(D.win_bad, winbad.data(), sizeof(int) * (size_t)bt->W, hipMemcpyHostToDevice));
    return VILO_OK;
  }
  if (!offsets) return VILO_ERR_BAD_ARG;
  if (!bt->leg || !bt->d_pre) { ctx->err = "vilo_batch_set_samples: the batch has no IMU-leg preintegration records"; return VILO_ERR_UNSUPPORTED; }
  VILO_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)bt->W * 10;
  std::vector<unsigned char> skip(n);
  VILO_HIP(hipMemcpy(skip.data(), D.imu_skip, n, hipMemcpyDeviceToHost));
  if (offsets[0] < 0) return VILO_ERR_BAD_ARG;
  for (size_t f = 0; f < n; ++f)
    if (offsets[f + 1] < offsets[f] || (!skip[f] && offsets[f + 1] - offsets[f] < 2)) {
      // (one sample = the constructor's: nothing is integrated and the covariance stays zero, which has no sqrt_info)
      ctx->err = "vilo_batch_set_samples: a live interval needs at least two samples";
      return VILO_ERR_BAD_ARG;
    }
  if (!bt->rp_s || bt->rp_cap < (size_t)offsets[n]) {
    bt->rp_cap = (size_t)offsets[n];
    int rc = dev_alloc(ctx, bt, &bt->rp_s, bt->rp_cap);
    if (rc == VILO_OK && !bt->rp_o) rc = dev_alloc(ctx, bt, &bt->rp_o, n + 1);
    if (rc == VILO_OK) rc = dev_alloc(ctx, bt, &bt->rp_t, bt->rp_cap * (size_t)(4 * VILO_LEG_REC));   // 4 legs x one record per sample
    if (rc != VILO_OK) {   // (nothing of a previous set of samples may stay referenced by the launch sequence)
      bt->rp_s = nullptr; bt->rp_cap = 0;
      D.rp_on = 0; D.rp_samples = nullptr; D.rp_terms = nullptr; D.rp_offsets = nullptr;
      if (bt->gexec) { (void)hipGraphExecDestroy(bt->gexec); bt->gexec = nullptr; }
      return rc;
    }
  }
  if (!bt->rp_orig) {   // the records the batch was created with (vilo_batch_set_samples(NULL) brings them back)
    int rc = dev_alloc(ctx, bt, &bt->rp_orig, n);
    if (rc != VILO_OK) return rc;
    VILO_HIP(hipMemcpyAsync(bt->rp_orig, bt->d_pre, sizeof(vilo_preint) * n, hipMemcpyDeviceToDevice, ctx->stream));
    VILO_HIP(hipStreamSynchronize(ctx->stream));
  }
  vilo_sample *d_s = bt->rp_s;
  int *d_o = bt->rp_o;
  double *d_t = bt->rp_t;
  // the records are integrated again before every linearisation from here on: the flags of the records the batch was created with no
  // longer describe them (k_accept / the marginalisation read prep_bad of the records in force)
  if (bt->d.win_bad) VILO_HIP(hipMemset(bt->d.win_bad, 0, sizeof(int) * (size_t)bt->W));
  VILO_HIP(hipMemcpy(d_s, samples, sizeof(vilo_sample) * (size_t)offsets[n], hipMemcpyHostToDevice));
  VILO_HIP(hipMemcpy(d_o, offsets, sizeof(int) * (n + 1), hipMemcpyHostToDevice));
  D.rp_samples = d_s; D.rp_terms = d_t; D.rp_offsets = d_o; D.rp_pre = bt->d_pre; D.prep_bad = bt->d_prep_bad; D.leg = 1; D.rp_on = 1;
  // Every interval is one IMULegIntegrationBase object that integrated its samples once (the record the batch was created with); the
  // solver's re-integrations are repropagate() calls on it, and repropagate() leaves the contact-force filter of contact_sensor_type 2 as
  // the previous pass left it (imu_leg_integration_base.cpp:62-86). The filter state after the original integration depends on the
  // samples only: that pass is run here once (mode 2: at the records' own linearisation point, which reproduces the records).
  if (!bt->rp_ff) { int rc = dev_alloc(ctx, bt, &bt->rp_ff, n * VILO_FF_N); if (rc != VILO_OK) { D.rp_on = 0; return rc; } }
  D.rp_ff = bt->rp_ff;
  VILO_HIP(hipMemsetAsync(bt->rp_ff, 0, sizeof(double) * n * VILO_FF_N, ctx->stream));
  if (ctx->cfg.contact_sensor_type == 2) {
    int rc = vilo_repropagate_launch(ctx, D, 2, 0);
    if (rc != VILO_OK) { D.rp_on = 0; return rc; }
    // a reset batch is the batch as it was after this call: the filter state the first integration left, not the one the last solve's
    // re-integrations left (reset + solve is then repeatable bit for bit: bench steps, Monte-Carlo restarts, graph replays)
    if (!bt->rp_ff0) { rc = dev_alloc(ctx, bt, &bt->rp_ff0, n * VILO_FF_N); if (rc != VILO_OK) { D.rp_on = 0; return rc; } }
    VILO_HIP(hipMemcpyAsync(bt->rp_ff0, bt->rp_ff, sizeof(double) * n * VILO_FF_N, hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (bt->gexec) { (void)hipGraphExecDestroy(bt->gexec); bt->gexec = nullptr; }   // the captured launch sequence changes
  return VILO_OK;
}

extern "C" int vilo_batch_reset(vilo_ctx *ctx, vilo_batch *bt) {
  if (!ctx || !bt) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  VILO_HIP(hipMemcpyAsync(bt->d.x, bt->d.x0, sizeof(double) * (size_t)bt->W * XSTRIDE, hipMemcpyDeviceToDevice, ctx->stream));
  if (bt->d.n_lm > 0)
    VILO_HIP(hipMemcpyAsync(bt->d.lam, bt->d.lam0, sizeof(double) * (size_t)bt->d.n_lm, hipMemcpyDeviceToDevice, ctx->stream));
  if (bt->d.rp_on && bt->rp_ff0 && ctx->cfg.contact_sensor_type == 2)   // the contact-force filters of the intervals' objects as set_samples left them
    VILO_HIP(hipMemcpyAsync(bt->rp_ff, bt->rp_ff0, sizeof(double) * (size_t)bt->W * 10 * VILO_FF_N, hipMemcpyDeviceToDevice, ctx->stream));
  return VILO_OK;
}

extern "C" int vilo_batch_solve(vilo_ctx *ctx, vilo_batch *bt, const vilo_solve_opts *opts) {
  if (!ctx || !bt || !opts) return VILO_ERR_BAD_ARG;
  if (opts->max_num_iterations < 0 || opts->max_num_iterations > 63) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  int rc = VILO_OK;
  const bool want_graph = !ctx->profile && !bt->graph_failed && bt->n_solves >= 1 && getenv("VILO_NO_GRAPH") == nullptr;
  if (opts->max_solver_time_us < 0) { ctx->err = "vilo_solve_opts.max_solver_time_us < 0 (fill the struct with vilo_default_solve_opts)"; return VILO_ERR_BAD_ARG; }
  if (want_graph && (!bt->gexec || memcmp(&bt->gopts, opts, sizeof(*opts)) != 0 || bt->g_sqrt_info_mode != ctx->sqrt_info_mode || bt->g_rp_on != bt->d.rp_on ||
                     bt->g_initial_mu != ctx->initial_mu || bt->g_solver_form != ctx->solver_form)) {
    if (bt->gexec) { (void)hipGraphExecDestroy(bt->gexec); bt->gexec = nullptr; }
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      rc = vilo_solve_launch(ctx, bt->d, opts);
      const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
      if (rc != VILO_OK || e != hipSuccess || !g || hipGraphInstantiate(&bt->gexec, g, nullptr, nullptr, 0) != hipSuccess) bt->gexec = nullptr;
      if (g) (void)hipGraphDestroy(g);
    }
    if (!bt->gexec) { bt->graph_failed = true; (void)hipGetLastError(); ctx->err.clear(); }
    else { bt->gopts = *opts; bt->g_sqrt_info_mode = ctx->sqrt_info_mode; bt->g_rp_on = bt->d.rp_on; bt->g_initial_mu = ctx->initial_mu; bt->g_solver_form = ctx->solver_form; }
    rc = VILO_OK;
  }
  VILO_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  if (want_graph && bt->gexec) {
    VILO_HIP(hipGraphLaunch(bt->gexec, ctx->stream));
  } else {
    rc = vilo_solve_launch(ctx, bt->d, opts);
    if (rc != VILO_OK) return rc;
  }
  ++bt->n_solves;
  VILO_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  VILO_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  VILO_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->last_solve_ms = ms;
  if (ctx->prep_pending) {   // (recorded on this stream before ev1: complete)
    float t = 0.f;
    if (hipEventElapsedTime(&t, ctx->prep_ev[0], ctx->prep_ev[1]) == hipSuccess) { ctx->kernel_ms[11] += t; ctx->kernel_launches[11] += 1; }
    ctx->prep_pending = false;
  }
  if (ctx->profile) {
    for (size_t i = 0; i < ctx->pev_kind.size(); ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ctx->pev[2 * i], ctx->pev[2 * i + 1]) == hipSuccess) {
        ctx->kernel_ms[ctx->pev_kind[i]] += t;
        ctx->kernel_launches[ctx->pev_kind[i]] += 1;
      }
    }
  }
  return VILO_OK;
}

extern "C" int vilo_batch_download(vilo_ctx *ctx, vilo_batch *bt, vilo_window_state *out, vilo_solve_summary *summ) {
  if (!ctx || !bt) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  const int W = bt->W;
  const auto t_dl0 = std::chrono::steady_clock::now();
  struct DlTimer { vilo_ctx *c; std::chrono::steady_clock::time_point t0; ~DlTimer() { c->last_download_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } dl_timer{ctx, t_dl0};
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  if (out) {
    std::vector<double> x((size_t)W * XSTRIDE), lam((size_t)std::max(1, bt->d.n_lm));
    VILO_HIP(hipMemcpy(x.data(), bt->d.x, sizeof(double) * x.size(), hipMemcpyDeviceToHost));
    if (bt->d.n_lm > 0) VILO_HIP(hipMemcpy(lam.data(), bt->d.lam, sizeof(double) * (size_t)bt->d.n_lm, hipMemcpyDeviceToHost));
    std::vector<WinMeta> wins(W);
    VILO_HIP(hipMemcpy(wins.data(), bt->d.win, sizeof(WinMeta) * W, hipMemcpyDeviceToHost));
    for (int w = 0; w < W; ++w) {
      const double *xw = &x[(size_t)w * XSTRIDE];
      const int F = wins[w].n_frames;
      memcpy(out[w].pose, xw + XO_POSE, sizeof(double) * 7 * F);
      memcpy(out[w].speed_bias, xw + XO_SB, sizeof(double) * 9 * F);
      memcpy(out[w].leg_bias, xw + XO_LB, sizeof(double) * 4 * F);
      memcpy(out[w].ex_pose, xw + XO_EX, sizeof(double) * 14);
      out[w].td[0] = xw[XO_TD];
      for (int i = 0; i < bt->L_host[w]; ++i) out[w].inv_depth[bt->perm_host[bt->lm_off_host[w] + i]] = lam[bt->lm_off_host[w] + i];
    }
  }
  if (summ) {
    std::vector<SolverState> st(W);
    VILO_HIP(hipMemcpy(st.data(), bt->d.st, sizeof(SolverState) * W, hipMemcpyDeviceToHost));
    for (int w = 0; w < W; ++w) {
      vilo_solve_summary &s = summ[w];
      memset(&s, 0, sizeof(s));
      s.iterations = st[w].iter; s.num_successful = st[w].num_successful; s.termination = st[w].termination;
      s.initial_cost = st[w].cost_trace[0]; s.final_cost = st[w].x_cost;
      memcpy(s.cost_trace, st[w].cost_trace, sizeof(s.cost_trace));
      memcpy(s.radius_trace, st[w].radius_trace, sizeof(s.radius_trace));
    }
  }
  return VILO_OK;
}

// one batch of host windows through create / solve / download on context c
static int solve_host_batch(vilo_ctx *c, int n, const vilo_window_desc *in, vilo_window_state *inout, const vilo_solve_opts *opts, vilo_solve_summary *out) {
  vilo_batch *bt = nullptr;
  int rc = vilo_batch_create(c, n, in, inout, &bt);
  if (rc != VILO_OK) return rc;
  rc = vilo_batch_solve(c, bt, opts);
  if (rc == VILO_OK) rc = vilo_batch_download(c, bt, inout, out);
  if (rc == VILO_OK && out) {
    for (int w = 0; w < n; ++w)
      if (out[w].termination == 2) rc = VILO_ERR_NUMERIC;
  }
  vilo_batch_destroy(c, bt);
  return rc;
}
static void snapshot_host_states(int n, const vilo_window_desc *in, const vilo_window_state *st, std::vector<double> &keep) {
  size_t tot = 0;
  for (int w = 0; w < n; ++w) tot += (size_t)20 * in[w].n_frames + 15 + (size_t)std::max(0, in[w].n_landmarks);
  keep.resize(tot);
  double *q = keep.data();
  for (int w = 0; w < n; ++w) {
    const int F = in[w].n_frames, L = std::max(0, in[w].n_landmarks);
    const vilo_window_state &s = st[w];
    memcpy(q, s.pose, sizeof(double) * 7 * F); q += 7 * F;
    memcpy(q, s.speed_bias, sizeof(double) * 9 * F); q += 9 * F;
    memcpy(q, s.leg_bias, sizeof(double) * 4 * F); q += 4 * F;
    memcpy(q, s.ex_pose, sizeof(double) * 14); q += 14;
    *q++ = s.td[0];
    if (L > 0) { memcpy(q, s.inv_depth, sizeof(double) * L); q += L; }
  }
}
static void restore_host_states(int n, const vilo_window_desc *in, vilo_window_state *inout, const std::vector<double> &keep) {
  const double *q = keep.data();
  for (int w = 0; w < n; ++w) {
    const int F = in[w].n_frames, L = std::max(0, in[w].n_landmarks);
    vilo_window_state &s = inout[w];
    memcpy(s.pose, q, sizeof(double) * 7 * F); q += 7 * F;
    memcpy(s.speed_bias, q, sizeof(double) * 9 * F); q += 9 * F;
    memcpy(s.leg_bias, q, sizeof(double) * 4 * F); q += 4 * F;
    memcpy(s.ex_pose, q, sizeof(double) * 14); q += 14;
    s.td[0] = *q++;
    if (L > 0) { memcpy(s.inv_depth, q, sizeof(double) * L); q += L; }
  }
}

// A call on many HOST windows (vilo_solve_windows, vilo_optimize_windows), cut into sub-batches that go through `lanes` internal contexts
// of the same device, one host thread each (vilo_set_host_pipeline; default 4 lanes of 1024 windows, from two sub-batches' worth of windows
// up): while one lane's windows are on the DMA engines or in the solver, the other lanes pack theirs — the three resources a hand-over of
// host windows needs (host cores, PCIe, GPU) work at the same time instead of one after the other. The windows are independent and every
// lane solves with the form the whole call would take as ONE batch, so the answer is the one batch's bit for bit
// (tests/test_gpu_parity.py::test_host_pipeline_...). `inout` comes back changed only if every sub-batch came through (or failed
// numerically, which is a per-window outcome): the states a finished sub-batch overwrote are put back.
// Returns false when the call is not one to cut (few windows, pipeline off, per-kernel profiling on): the caller runs it as one batch.
bool vilo_run_on_lanes(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, vilo_window_state *inout,
                       const std::function<int(vilo_ctx *lane, int w0, int n)> &fn, int *rc_out) {
  const int sub = ctx->pipe_sub, lanes_want = ctx->pipe_lanes;
  if (lanes_want < 2 || sub <= 0 || n_windows < 2 * sub || ctx->profile) return false;
  for (int w = 0; w < n_windows; ++w) {   // (what the snapshots below index with; the sub-batches' own checks say why)
    const vilo_window_state &s = inout[w];
    if (in[w].n_frames < 2 || in[w].n_frames > VILO_MAX_FRAMES || in[w].n_landmarks < 0 || in[w].n_landmarks > VILO_NUM_OF_F || !s.pose || !s.speed_bias || !s.leg_bias ||
        !s.ex_pose || !s.td || (in[w].n_landmarks && !s.inv_depth)) return false;
  }
  // The kernel set of a batch depends on its size (small assembly up to 256 windows, frame-parallel visual form up to 256 packed waves,
  // compact rows while every window keeps td constant), and those forms agree to rounding, not bitwise. The lanes run the kernel set of a
  // FULL batch whatever their share (BatchDev::full_regime) with the row form of the whole call, so a call is cut only if as ONE batch it
  // would certainly be a full one too: more windows than the small assembly takes, more windows with landmarks (>= packed waves) than the
  // frame-parallel form takes.
  int with_lm = 0, all_td_const = 1;
  for (int w = 0; w < n_windows; ++w) { with_lm += in[w].n_landmarks > 0 ? 1 : 0; if (!in[w].td_const) all_td_const = 0; }
  if (n_windows <= vilo_assemble_small_max() || (size_t)with_lm <= vilo_tpar_max_waves()) return false;
  const int n_sub = (n_windows + sub - 1) / sub, per = (n_windows + n_sub - 1) / n_sub;   // equal shares: no small tail batch
  const int n_lanes = std::min(lanes_want, n_sub);
  while ((int)ctx->lanes.size() < n_lanes) {
    vilo_ctx *l = nullptr;
    const int rc = vilo_create(&l, &ctx->cfg, ctx->device);
    if (rc != VILO_OK) { ctx->err = "no context for a pipeline lane"; *rc_out = rc; return true; }
    // the lanes pack side by side: each brings its own share of the host's threads (at most the shared pool's 16, at least 2)
    const int hw = (int)std::thread::hardware_concurrency();
    l->dma_turn = &ctx->dma_m;
    l->pool = new vilo::WorkerPool(std::max(2, std::min(16, (hw > 0 ? hw : 1) / std::max(1, lanes_want))) - 1);
    ctx->lanes.push_back(l);
  }
  BatchDev probe;
  memset(&probe, 0, sizeof(probe));
  probe.W = n_windows;
  const int form = vilo_solver_form(ctx, probe);   // what ONE batch of all the windows would be solved with
  std::vector<int> rcs(n_sub, VILO_OK);
  std::vector<char> ran(n_sub, 0);
  std::vector<std::vector<double>> keep(n_sub);
  std::vector<double> solve_ms(n_lanes, 0.0), marg_ms(n_lanes, 0.0);
  std::vector<int> general(n_lanes, 0);
  std::vector<std::thread> th;
  for (int li = 0; li < n_lanes; ++li) {
    vilo_ctx *l = ctx->lanes[li];
    l->sqrt_info_mode = ctx->sqrt_info_mode; l->solver_form = form; l->compact_rows = (ctx->compact_rows && all_td_const) ? 1 : 0; l->regime_full = 1;
    l->initial_mu = ctx->initial_mu; l->prior_form = ctx->prior_form; l->err.clear();
  }
  auto lane_work = [&](int li) {
    vilo_ctx *l = ctx->lanes[li];
    for (int i = li; i < n_sub; i += n_lanes) {
      const int w0 = i * per, n = std::min(per, n_windows - w0);
      if (n <= 0) break;
      try {
        snapshot_host_states(n, in + w0, inout + w0, keep[i]);
        l->last_solve_ms = 0.0; l->last_marg_ms = 0.0; l->marg_general_count = 0;
        ran[i] = 1;
        rcs[i] = fn(l, w0, n);
      } catch (...) {   // (host allocation failure inside a lane's thread: reported, not thrown across the C boundary)
        rcs[i] = VILO_ERR_HIP; l->err = "host allocation failed in a pipeline lane";
      }
      solve_ms[li] += l->last_solve_ms; marg_ms[li] += l->last_marg_ms; general[li] += l->marg_general_count;
      if (rcs[i] != VILO_OK && rcs[i] != VILO_ERR_NUMERIC) break;
    }
  };
  std::vector<int> inline_lanes;   // (a lane whose thread the process could not start is worked on this one)
  for (int li = 0; li < n_lanes; ++li) {
    try { th.emplace_back(lane_work, li); } catch (...) { inline_lanes.push_back(li); }
  }
  for (int li : inline_lanes) lane_work(li);
  for (std::thread &t : th) t.join();
  int rc = VILO_OK;
  for (int i = 0; i < n_sub; ++i)
    if (rcs[i] != VILO_OK && rcs[i] != VILO_ERR_NUMERIC) { rc = rcs[i]; ctx->err = ctx->lanes[i % n_lanes]->err; break; }
  if (rc != VILO_OK) {   // a sub-batch was refused or lost its device: the call as a whole did not happen
    for (int i = 0; i < n_sub; ++i) {
      const int w0 = i * per, n = std::min(per, n_windows - w0);
      if (n > 0 && ran[i]) restore_host_states(n, in + w0, inout + w0, keep[i]);
    }
    *rc_out = rc;
    return true;
  }
  ctx->last_solve_ms = 0.0; ctx->last_marg_ms = 0.0; ctx->marg_general_count = 0;
  for (int li = 0; li < n_lanes; ++li) { ctx->last_solve_ms += solve_ms[li]; ctx->last_marg_ms += marg_ms[li]; ctx->marg_general_count += general[li]; }
  for (int i = 0; i < n_sub; ++i) if (rcs[i] == VILO_ERR_NUMERIC) rc = VILO_ERR_NUMERIC;
  *rc_out = rc;
  return true;
}

// Estimator::optimization()'s solve half on host windows (estimator.cpp:1054-1245)
extern "C" int vilo_solve_windows(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, vilo_window_state *inout,
                                  const vilo_solve_opts *opts, vilo_solve_summary *out) {
  if (!ctx || !in || !inout || !opts || n_windows <= 0) return VILO_ERR_BAD_ARG;
  int rc = VILO_OK;
  if (vilo_run_on_lanes(ctx, n_windows, in, inout, [&](vilo_ctx *l, int w0, int n) { return solve_host_batch(l, n, in + w0, inout + w0, opts, out ? out + w0 : nullptr); }, &rc))
    return rc;
  return solve_host_batch(ctx, n_windows, in, inout, opts, out);
}

// lanes < 2 or sub_windows <= 0: every vilo_solve_windows call is one batch
extern "C" int vilo_set_host_pipeline(vilo_ctx *ctx, int lanes, int sub_windows) {
  if (!ctx || lanes < 0 || lanes > 8 || sub_windows < 0) return VILO_ERR_BAD_ARG;
  ctx->pipe_lanes = lanes; ctx->pipe_sub = sub_windows;
  return VILO_OK;
}

// host wall time inside the last vilo_batch_create of this context: [0] total, [1] packing, [2] allocation + upload of observations /
// states / priors, [3] preintegration records up + sqrt_info; and the bytes it moved to the device
extern "C" int vilo_last_create_ms(const vilo_ctx *ctx, double out_ms[4], double *bytes_up) {
  if (!ctx || !out_ms) return VILO_ERR_BAD_ARG;
  for (int i = 0; i < 4; ++i) out_ms[i] = ctx->last_create_ms[i];
  if (bytes_up) *bytes_up = ctx->last_create_bytes;
  return VILO_OK;
}
extern "C" double vilo_last_download_ms(const vilo_ctx *ctx) { return ctx ? ctx->last_download_ms : -1.0; }

// Test / profiling hook: copy an internal device array of window `win` to the host. Not part of the
// reference's interface. what: 0 gram slots, 1 lm_E, 2 lm_g, 3 lm_w (80 x L), 4 cam_g, 5 cam_dh2, 6 cam_y,
// 7 imu_lin, 8 lm_y, 9 lm_dh2, 10 SolverState scalars + cost trace (24 + 64 doubles), 11 landmark permutation (as doubles),
// 13 the preintegration records (10 x vilo_preint)
extern "C" int vilo_debug_fetch(vilo_ctx *ctx, vilo_batch *bt, int what, int win, double *out, int max_n) {
  if (!ctx || !bt || win < 0 || win >= bt->W || !out) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  WinMeta wm;
  VILO_HIP(hipMemcpy(&wm, bt->d.win + win, sizeof(WinMeta), hipMemcpyDeviceToHost));
  const double *src = nullptr;
  size_t n = 0;
  switch (what) {
    case 0: src = bt->d.gram + (size_t)wm.gram_off * VILO_GRAM; n = (size_t)wm.n_gram * VILO_GRAM; break;
    case 1: src = bt->d.lm_E + wm.lm_off; n = wm.L; break;
    case 2: {
      SolverState sst;
      VILO_HIP(hipMemcpy(&sst, bt->d.st + win, sizeof(SolverState), hipMemcpyDeviceToHost));
      src = bt->d.lm_gbuf[sst.cur & 1] + wm.lm_off; n = wm.L; break;
    }
    case 3: src = bt->d.lm_w + 80 * (size_t)wm.lm_off; n = (size_t)80 * wm.L; break;
    case 4: src = bt->d.cam_g + (size_t)win * CD_N; n = CD_N; break;
    case 5: src = bt->d.cam_dh2 + (size_t)win * CD_N; n = CD_N; break;
    case 6: src = bt->d.cam_y + (size_t)win * CD_N; n = CD_N; break;
    case 7: src = bt->d.imu_lin + (size_t)win * 10 * 31 * 39; n = 10 * 31 * 39; break;
    case 8: src = bt->d.lm_y + wm.lm_off; n = wm.L; break;
    case 9: src = bt->d.lm_dh2 + wm.lm_off; n = wm.L; break;
    case 10: src = (const double *)(bt->d.st + win); n = 24 + 64; break;
    case 12: src = (const double *)(bt->d.st + win) + 24 + 128 + 1; n = 63; break;   // phase_clk (int64 bit patterns)
    case 13:   // the window's preintegration records as they stand (vilo_preint x 10; with vilo_batch_set_samples: the last re-integration)
      if (!bt->leg || !bt->d_pre) return VILO_ERR_UNSUPPORTED;
      src = (const double *)((const vilo_preint *)bt->d_pre + (size_t)win * 10); n = 10 * sizeof(vilo_preint) / sizeof(double); break;
    case 11: {
      n = wm.L;
      if ((int)n > max_n) return VILO_ERR_BAD_ARG;
      for (size_t i = 0; i < n; ++i) out[i] = bt->perm_host[bt->lm_off_host[win] + i];
      return (int)n;
    }
    default: return VILO_ERR_BAD_ARG;
  }
  if ((long long)n > max_n) return VILO_ERR_BAD_ARG;
  if (n) VILO_HIP(hipMemcpy(out, src, n * sizeof(double), hipMemcpyDeviceToHost));
  return (int)n;
}
