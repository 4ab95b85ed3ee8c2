// Context management of libvilo_gpu.so. There is deliberately NO CPU fallback anywhere in this library:
// without a usable HIP device vilo_create fails with VILO_ERR_NO_DEVICE and nothing else can be called.
#include "vilo_internal.hpp"

extern "C" void vilo_default_config(vilo_config *c) {
  // config/a1_config/hardware_a1_vilo_config.yaml:24-48,85-99 ; estimator.cpp:140-163 ; parameters.h:22
  memset(c, 0, sizeof(*c));
  c->acc_n = 0.9; c->acc_n_z = 2.5; c->acc_w = 0.0004; c->gyr_n = 0.05; c->gyr_w = 0.0002;
  c->g_norm = 9.805;
  c->phi_n = 0.00001; c->dphi_n = 0.00001;
  c->rho_c_n = 0.00000001; c->rho_nc_n = 0.00000000001;
  c->v_n_min_xy = 0.001; c->v_n_min_z = 0.005; c->v_n_min = 0.005; c->v_n_max = 900.0;
  c->v_n_force_thres_ratio = 0.8; c->v_n_term1_steep = 10; c->v_n_term2_var_rescale = 1.0e-6;
  c->v_n_term3_distance_rescale = 1.0e-3;
  c->contact_sensor_type = 0;
  const double ox[4] = {0.1805, 0.1805, -0.1805, -0.1805};
  const double oy[4] = {0.047, -0.047, 0.047, -0.047};
  const double d[4] = {0.0838, -0.0838, 0.0838, -0.0838};
  for (int j = 0; j < 4; ++j) {
    c->rho_fix[j][0] = ox[j]; c->rho_fix[j][1] = oy[j]; c->rho_fix[j][2] = d[j]; c->rho_fix[j][3] = 0.21;
  }
  c->R_br[0] = c->R_br[4] = c->R_br[8] = 1.0;
  c->focal_length = 460.0;
  c->huber_delta = 1.0;
}

extern "C" void vilo_default_solve_opts(vilo_solve_opts *o) {
  // estimator.cpp:1221-1233 + Ceres 1.14 Solver::Options defaults
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 12;
  o->fixed_iterations = 0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1;
}

extern "C" int vilo_create(vilo_ctx **out, const vilo_config *cfg, int device) {
  if (!out || !cfg) return VILO_ERR_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return VILO_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return VILO_ERR_NO_DEVICE;
  vilo_ctx *ctx = new vilo_ctx();
  ctx->cfg = *cfg;
  ctx->device = device;
  ctx->last_solve_ms = 0.0;
  ctx->d_cfg = nullptr;
  ctx->profile = 0;
  if (const char *e = getenv("VILO_SOLVER"))
    ctx->solver_form = !strcmp(e, "mw8") ? VILO_SOLVER_MW8 : (!strcmp(e, "wave") ? VILO_SOLVER_WAVE : (!strcmp(e, "split") ? VILO_SOLVER_SPLIT : VILO_SOLVER_AUTO));
  ctx->compact_rows = getenv("VILO_NO_COMPACT") ? 0 : 1;
  if (const char *e = getenv("VILO_HOST_PIPELINE")) {   // "lanes,sub_windows"; "0" switches the sub-batch pipeline of vilo_solve_windows off
    int l = 0, sw = 1024;
    if (sscanf(e, "%d,%d", &l, &sw) >= 1 && l >= 0 && l <= 8 && sw >= 0) { ctx->pipe_lanes = l; ctx->pipe_sub = sw; }
  }
  for (int i = 0; i < VILO_NKERNEL; ++i) { ctx->kernel_ms[i] = 0.0; ctx->kernel_launches[i] = 0; }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
      hipEventCreate(&ctx->ev1) != hipSuccess || hipMalloc((void **)&ctx->d_cfg, sizeof(vilo_config)) != hipSuccess ||
      hipMemcpy(ctx->d_cfg, cfg, sizeof(vilo_config), hipMemcpyHostToDevice) != hipSuccess) {
    delete ctx;
    return VILO_ERR_HIP;
  }
  *out = ctx;
  return VILO_OK;
}

extern "C" int vilo_device_count(void) {
  int ndev = 0;
  return (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) ? ndev : 0;
}

extern "C" void vilo_destroy(vilo_ctx *ctx) {
  if (!ctx) return;
  for (vilo_ctx *l : ctx->lanes) vilo_destroy(l);
  ctx->lanes.clear();
  if (ctx->pool) { ctx->pool->shutdown(); delete ctx->pool; ctx->pool = nullptr; }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_cfg) (void)hipFree(ctx->d_cfg);
  for (auto &c : ctx->pool_free) (void)hipFree(c.first);
  for (auto &h : ctx->host_stage) if (h.first) { if (h.second & 1) (void)hipHostFree(h.first); else free(h.first); }
  for (hipEvent_t e : ctx->pev) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->prep_ev) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->rec_ev) if (e) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char *vilo_last_error(const vilo_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" double vilo_last_solve_ms(const vilo_ctx *ctx) { return ctx ? ctx->last_solve_ms : -1.0; }

// Profiling hooks (not part of the reference interface): per-kernel GPU time of the solve pipeline measured with
// HIP events on the stream the kernels are launched on.
extern "C" int vilo_set_sqrt_info_mode(vilo_ctx *ctx, int mode) {
  if (!ctx || (mode != 0 && mode != 1)) return VILO_ERR_BAD_ARG;
  ctx->sqrt_info_mode = mode;
  return VILO_OK;
}
// Form of the square root a marginalisation leaves (include/vilo_gpu.h).
extern "C" int vilo_set_prior_form(vilo_ctx *ctx, int form) {
  if (!ctx || (form != VILO_PRIOR_EIGEN && form != VILO_PRIOR_FACTOR)) return VILO_ERR_BAD_ARG;
  ctx->prior_form = form;
  return VILO_OK;
}
// Which of the solver's forms the batches of this context take. The forms restate one algorithm; the multi-wave ones differ from the
// single wave in elimination order and agree with it to rounding, not bitwise, so a caller that needs the same answer for a window
// whatever the size of the batch it shares pins one (the choice is part of the key of a batch's captured launch sequence).
extern "C" int vilo_set_solver_form(vilo_ctx *ctx, int form) {
  if (!ctx || !(form == VILO_SOLVER_AUTO || form == VILO_SOLVER_WAVE || form == VILO_SOLVER_SPLIT || form == VILO_SOLVER_MW8)) return VILO_ERR_BAD_ARG;
  ctx->solver_form = form;
  return VILO_OK;
}
extern "C" int vilo_get_solver_form(const vilo_ctx *ctx) { return ctx ? ctx->solver_form : VILO_ERR_BAD_ARG; }
// Batches created from here on may (1, default) or may not (0) use the compact 16-column visual rows when every window keeps td constant.
extern "C" int vilo_set_compact_rows(vilo_ctx *ctx, int on) {
  if (!ctx || (on != 0 && on != 1)) return VILO_ERR_BAD_ARG;
  ctx->compact_rows = on;
  return VILO_OK;
}
// Test hook (not part of the reference's interface): DoglegStrategy's mu at the start of the next solves, so that a solve can be resumed
// from a state (x, radius, mu) another implementation reached — tests/test_branches.py compares single steps with the oracle.
extern "C" int vilo_debug_set_initial_mu(vilo_ctx *ctx, double mu) {
  if (!ctx || !(mu > 0.0)) return VILO_ERR_BAD_ARG;
  ctx->initial_mu = mu;
  return VILO_OK;
}
extern "C" void vilo_set_profiling(vilo_ctx *ctx, int on) {
  if (!ctx) return;
  ctx->profile = on;
  for (int i = 0; i < VILO_NKERNEL; ++i) { ctx->kernel_ms[i] = 0.0; ctx->kernel_launches[i] = 0; }
}
extern "C" int vilo_get_kernel_times(const vilo_ctx *ctx, double *ms, long long *launches, int n) {
  if (!ctx || !ms || !launches) return VILO_ERR_BAD_ARG;
  for (int i = 0; i < n && i < VILO_NKERNEL; ++i) { ms[i] = ctx->kernel_ms[i]; launches[i] = ctx->kernel_launches[i]; }
  return VILO_NKERNEL;
}
extern "C" const char *vilo_kernel_name(int kind) {
  static const char *names[VILO_NKERNEL] = {"k_visual_linearize", "k_imu_linearize", "k_assemble_bias", "k_visual_cost", "k_imu_cost", "k_accept", "k_init_state", "k_imu_raw", "k_assemble", "k_solve_wave", "k_repropagate", "k_prepare_preint", "k_chain", "k_solve_mid", "k_backsub"};
  return (kind >= 0 && kind < VILO_NKERNEL) ? names[kind] : "";
}
