// Deterministic synthetic sliding-window generator (see include/vilo_synth.h).
// Host-only tooling: it produces the *inputs* of Estimator::optimization() that the reference receives
// from ROS (src/main.cpp:255-393), the feature manager (feature_manager.h:28-78) and
// Estimator::processIMULeg (estimator.cpp:590-653). No oracle code, no GPU.
#include <string.h>

#include <map>
#include <set>
#include <vector>

#include "../../include/vilo_synth.h"
#include "../csrc/vilo_math.hpp"

using namespace vilo;

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
  double normal() {
    double u1 = uni(), u2 = uni();
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
  }
};

const double TWO_PI = 6.283185307179586;

// trot-like forward motion 0.5 m/s + sinusoids
void traj(double t, v3 &p, v3 &v, v3 &a, double ypr[3], double dypr[3]) {
  const double w1 = TWO_PI * 1.3, w2 = TWO_PI * 0.7, w3 = TWO_PI * 2.5;
  p = mk3(0.5 * t + 0.02 * sin(w1 * t), 0.03 * sin(w2 * t), 0.30 + 0.01 * sin(w3 * t));
  v = mk3(0.5 + 0.02 * w1 * cos(w1 * t), 0.03 * w2 * cos(w2 * t), 0.01 * w3 * cos(w3 * t));
  a = mk3(-0.02 * w1 * w1 * sin(w1 * t), -0.03 * w2 * w2 * sin(w2 * t), -0.01 * w3 * w3 * sin(w3 * t));
  const double wy = TWO_PI * 0.3, wp = TWO_PI * 0.9, wr = TWO_PI * 1.25;
  ypr[0] = 0.10 * sin(wy * t); dypr[0] = 0.10 * wy * cos(wy * t);
  ypr[1] = 0.02 * sin(wp * t + 0.5); dypr[1] = 0.02 * wp * cos(wp * t + 0.5);
  ypr[2] = 0.03 * sin(wr * t); dypr[2] = 0.03 * wr * cos(wr * t);
}
m3 Rzyx(const double ypr[3]) {
  const double cy = cos(ypr[0]), sy = sin(ypr[0]), cp = cos(ypr[1]), sp = sin(ypr[1]), cr = cos(ypr[2]), sr = sin(ypr[2]);
  m3 R;
  R.a[0] = cy * cp; R.a[1] = cy * sp * sr - sy * cr; R.a[2] = cy * sp * cr + sy * sr;
  R.a[3] = sy * cp; R.a[4] = sy * sp * sr + cy * cr; R.a[5] = sy * sp * cr - cy * sr;
  R.a[6] = -sp; R.a[7] = cp * sr; R.a[8] = cp * cr;
  return R;
}
v3 body_omega(const double ypr[3], const double d[3]) {
  const double sp = sin(ypr[1]), cp = cos(ypr[1]), sr = sin(ypr[2]), cr = cos(ypr[2]);
  return mk3(d[2] - d[0] * sp, d[1] * cr + d[0] * sr * cp, -d[1] * sr + d[0] * cr * cp);
}
quat quat_from_R(const m3 &m) {
  quat q;
  double t = m.a[0] + m.a[4] + m.a[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m.a[7] - m.a[5]) * t; q.y = (m.a[2] - m.a[6]) * t; q.z = (m.a[3] - m.a[1]) * t;
  } else {
    int i = 0;
    if (m.a[4] > m.a[0]) i = 1;
    if (m.a[8] > m.a[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m.a[4 * i] - m.a[4 * j] - m.a[4 * k] + 1.0);
    double vv[3];
    vv[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.a[3 * k + j] - m.a[3 * j + k]) * t;
    vv[j] = (m.a[3 * j + i] + m.a[3 * i + j]) * t;
    vv[k] = (m.a[3 * k + i] + m.a[3 * i + k]) * t;
    q.x = vv[0]; q.y = vv[1]; q.z = vv[2];
  }
  return q;
}
m3 inv3(const m3 &A) {
  m3 o;
  const double *a = A.a;
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  const double id = 1.0 / det;
  o.a[0] = (a[4] * a[8] - a[5] * a[7]) * id; o.a[1] = (a[2] * a[7] - a[1] * a[8]) * id; o.a[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o.a[3] = (a[5] * a[6] - a[3] * a[8]) * id; o.a[4] = (a[0] * a[8] - a[2] * a[6]) * id; o.a[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o.a[6] = (a[3] * a[7] - a[4] * a[6]) * id; o.a[7] = (a[1] * a[6] - a[0] * a[7]) * id; o.a[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  return o;
}

struct BodyState {
  v3 p, v, a, w;
  m3 R;
};
BodyState body_at(double t) {
  BodyState b;
  double ypr[3], d[3];
  traj(t, b.p, b.v, b.a, ypr, d);
  b.R = Rzyx(ypr);
  b.w = body_omega(ypr, d);
  return b;
}

const double GAIT_T = 0.4;
// stance index of leg j at time t (integer id of the stance/swing half-period), and whether in stance
bool in_stance(int leg, double t, double &phase_start) {
  const bool groupA = (leg == 0 || leg == 3);
  double tt = t + 10 * GAIT_T;  // keep positive
  double ph = fmod(tt, GAIT_T);
  bool first_half = ph < 0.5 * GAIT_T;
  phase_start = t - (first_half ? ph : ph - 0.5 * GAIT_T);
  return groupA ? first_half : !first_half;
}
// world foothold used for the stance phase starting at ts
v3 foothold(const vilo_config *cfg, int leg, double ts) {
  BodyState mid = body_at(ts + 0.25 * GAIT_T);
  const double *rf = cfg->rho_fix[leg];
  double ypr[3] = {0, 0, 0}, d[3];
  v3 p, v, a;
  traj(ts + 0.25 * GAIT_T, p, v, a, ypr, d);
  double yaw_only[3] = {ypr[0], 0, 0};
  v3 w = mid.p + Rzyx(yaw_only) * mk3(rf[0], rf[1] + rf[2], 0.0);
  w.z = 0.0;
  return w;
}
// foot position / velocity in the body frame at time t
void foot_body(const vilo_config *cfg, int leg, double t, v3 &f, v3 &df, double &contact) {
  double ps;
  const bool st = in_stance(leg, t, ps);
  BodyState b = body_at(t);
  if (st) {
    v3 pf = foothold(cfg, leg, ps);
    f = tr(b.R) * (pf - b.p);
    df = -cross(b.w, f) - tr(b.R) * b.v;
    contact = 1.0;
  } else {
    const double t_lo = ps, t_td = ps + 0.5 * GAIT_T;
    BodyState blo = body_at(t_lo), btd = body_at(t_td);
    v3 f_lo = tr(blo.R) * (foothold(cfg, leg, ps - 0.5 * GAIT_T) - blo.p);
    v3 f_td = tr(btd.R) * (foothold(cfg, leg, t_td) - btd.p);
    const double Ts = 0.5 * GAIT_T, tau = (t - t_lo) / Ts;
    const double sm = tau * tau * (3 - 2 * tau), dsm = 6 * tau * (1 - tau) / Ts;
    f = f_lo + (f_td - f_lo) * sm;
    f.z += 0.05 * sin(M_PI * tau);
    df = (f_td - f_lo) * dsm;
    df.z += 0.05 * M_PI * cos(M_PI * tau) / Ts;
    contact = 0.0;
  }
}
void leg_ik(const double *rf, double lc, const v3 &target, double q[3]) {
  q[0] = 0.0; q[1] = 0.8; q[2] = -1.6;
  for (int it = 0; it < 20; ++it) {
    v3 f;
    m3 J;
    leg_fk_jac(q, lc, rf, f, J);
    v3 dq = inv3(J) * (target - f);
    q[0] += dq.x; q[1] += dq.y; q[2] += dq.z;
    if (norm(dq) < 1e-14) break;
  }
}

void make_sample(const vilo_config *cfg, double t, double dt, const double lc_true[4], const v3 &ba, const v3 &bg, Rng &rng,
                 vilo_sample *s) {
  BodyState b = body_at(t);
  const v3 g = mk3(0, 0, cfg->g_norm);
  v3 acc = tr(b.R) * (b.a + g) + ba + mk3(rng.normal(), rng.normal(), rng.normal()) * 0.1;
  v3 gyr = b.w + bg + mk3(rng.normal(), rng.normal(), rng.normal()) * 0.01;
  s->dt = dt;
  st3(s->acc, acc);
  st3(s->gyr, gyr);
  for (int j = 0; j < 4; ++j) {
    v3 f, df;
    double c;
    foot_body(cfg, j, t, f, df, c);
    double q[3];
    leg_ik(cfg->rho_fix[j], lc_true[j], f, q);
    v3 ff;
    m3 J;
    leg_fk_jac(q, lc_true[j], cfg->rho_fix[j], ff, J);
    v3 dq = inv3(J) * df;
    const double dqv[3] = {dq.x, dq.y, dq.z};
    for (int k = 0; k < 3; ++k) {
      s->phi[3 * j + k] = q[k] + 1e-4 * rng.normal();
      s->dphi[3 * j + k] = dqv[k] + 1e-3 * rng.normal();
    }
    s->c[j] = c;
  }
}

const int F = VILO_MAX_FRAMES;

void interval_layout(const vilo_synth_params *p, int &n_full, double &h, double &last_dt) {
  const double DT = 1.0 / p->frame_rate_hz;
  h = 1.0 / p->imu_rate_hz;
  n_full = (int)floor(DT / h + 1e-9);
  last_dt = DT - n_full * h;
  if (last_dt < 1e-9) { last_dt = h; n_full -= 1; }
}

}  // namespace

extern "C" void vilo_synth_default_params(vilo_synth_params *p, int config) {
  memset(p, 0, sizeof(*p));
  p->seed = 20260925ull;
  p->n_landmarks = (config == 3) ? 1000 : 200;
  p->n_start_frames = 7;
  p->imu_rate_hz = (config == 3) ? 400.0 : 500.0;
  p->frame_rate_hz = 15.0;
  p->pixel_noise = 0.5;
  p->sig_p = 0.02; p->sig_theta = 0.5 * M_PI / 180.0; p->sig_v = 0.05; p->sig_ba = 0.02; p->sig_bg = 0.002;
  p->sig_rho = 0.003; p->sig_lambda_rel = 0.10;
  p->lin_offset_ba = 0.005; p->lin_offset_bg = 0.0005; p->lin_offset_rho = 0.0005;
  p->with_prior = 1;
}

extern "C" void vilo_synth_sizes(const vilo_synth_params *p, int32_t *n_obs, int32_t *n_samples) {
  int no = 0;
  for (int i = 0; i < p->n_landmarks; ++i) no += F - (i % p->n_start_frames);
  int n_full;
  double h, last;
  interval_layout(p, n_full, h, last);
  *n_obs = no;
  *n_samples = (F - 1) * (n_full + 2);  // ctor sample + n_full full steps + 1 trimmed step
}

extern "C" int vilo_synth_window(const vilo_config *cfg, const vilo_synth_params *p, vilo_synth_out *o) {
  Rng rng(p->seed * 0x9E3779B97F4A7C15ull + 12345);
  const double DT = 1.0 / p->frame_rate_hz;
  const int L = p->n_landmarks;
  // extrinsics: config/a1_config/hardware_a1_vilo_config.yaml:55-71
  m3 ric;
  ric.a[0] = 0; ric.a[1] = 0; ric.a[2] = 1; ric.a[3] = -1; ric.a[4] = 0; ric.a[5] = 0; ric.a[6] = 0; ric.a[7] = -1; ric.a[8] = 0;
  const v3 tic0 = mk3(0.10076, 0.025, 0.1114), tic1 = mk3(0.10076, -0.025, 0.1114);
  const quat qic = quat_from_R(ric);
  const v3 ba_true = mk3(0.05, -0.03, 0.02), bg_true = mk3(0.002, -0.001, 0.0015);
  double lc_true[4];
  for (int j = 0; j < 4; ++j) lc_true[j] = 0.21 + 0.003 * rng.normal();

  // --- truth + initial states ---
  BodyState fs[F];
  for (int k = 0; k < F; ++k) {
    fs[k] = body_at(k * DT);
    quat q = quat_from_R(fs[k].R);
    double *tp = o->truth_pose + 7 * k, *ts = o->truth_speed_bias + 9 * k;
    st3(tp, fs[k].p);
    tp[3] = q.x; tp[4] = q.y; tp[5] = q.z; tp[6] = q.w;
    st3(ts, fs[k].v); st3(ts + 3, ba_true); st3(ts + 6, bg_true);
    for (int j = 0; j < 4; ++j) o->truth_leg_bias[4 * k + j] = lc_true[j];
    double d6[6];
    for (int i = 0; i < 3; ++i) d6[i] = p->sig_p * rng.normal();
    for (int i = 3; i < 6; ++i) d6[i] = p->sig_theta * rng.normal();
    pose_plus(tp, d6, o->pose + 7 * k);
    double *sb = o->speed_bias + 9 * k;
    for (int i = 0; i < 3; ++i) sb[i] = ts[i] + p->sig_v * rng.normal();
    for (int i = 3; i < 6; ++i) sb[i] = ts[i] + p->sig_ba * rng.normal();
    for (int i = 6; i < 9; ++i) sb[i] = ts[i] + p->sig_bg * rng.normal();
    for (int j = 0; j < 4; ++j) o->leg_bias[4 * k + j] = lc_true[j] + p->sig_rho * rng.normal();
  }
  for (int c = 0; c < 2; ++c) {
    double *e = o->ex_pose + 7 * c;
    st3(e, c == 0 ? tic0 : tic1);
    e[3] = qic.x; e[4] = qic.y; e[5] = qic.z; e[6] = qic.w;
  }
  o->td[0] = 0.0024;  // yaml:99

  // --- sensor samples per interval ---
  int n_full;
  double h, last_dt;
  interval_layout(p, n_full, h, last_dt);
  int so = 0;
  for (int k = 0; k + 1 < F; ++k) {
    o->sample_offsets[k] = so;
    const double t0 = k * DT;
    make_sample(cfg, t0, 0.0, lc_true, ba_true, bg_true, rng, &o->samples[so++]);
    for (int i = 1; i <= n_full; ++i) make_sample(cfg, t0 + i * h, h, lc_true, ba_true, bg_true, rng, &o->samples[so++]);
    make_sample(cfg, t0 + DT, last_dt, lc_true, ba_true, bg_true, rng, &o->samples[so++]);
    double *lin = o->lin + 10 * k;
    const double *sb = o->speed_bias + 9 * k;
    for (int i = 0; i < 3; ++i) lin[i] = sb[3 + i] + p->lin_offset_ba * rng.normal();
    for (int i = 0; i < 3; ++i) lin[3 + i] = sb[6 + i] + p->lin_offset_bg * rng.normal();
    for (int j = 0; j < 4; ++j) lin[6 + j] = o->leg_bias[4 * k + j] + p->lin_offset_rho * rng.normal();
  }
  o->sample_offsets[F - 1] = so;

  // --- landmarks and stereo observations ---
  const double pn = p->pixel_noise / cfg->focal_length;
  int oo = 0;
  std::vector<double> pl(2 * F), pr(2 * F);
  for (int i = 0; i < L; ++i) {
    const int s = i % p->n_start_frames;
    o->lm_start_frame[i] = s;
    o->lm_obs_offset[i] = oo;
    const double x = rng.uni(-0.6, 0.6), y = rng.uni(-0.6, 0.6), depth = rng.uni(2.0, 15.0);
    const v3 Pw = fs[s].R * (ric * (mk3(x, y, 1.0) * depth) + tic0) + fs[s].p;
    o->truth_inv_depth[i] = 1.0 / depth;
    o->inv_depth[i] = (1.0 / depth) * (1.0 + p->sig_lambda_rel * rng.normal());
    for (int j = s; j < F; ++j) {
      const v3 pb = tr(fs[j].R) * (Pw - fs[j].p);
      const v3 pc0 = tr(ric) * (pb - tic0), pc1 = tr(ric) * (pb - tic1);
      pl[2 * j] = pc0.x / pc0.z + pn * rng.normal(); pl[2 * j + 1] = pc0.y / pc0.z + pn * rng.normal();
      pr[2 * j] = pc1.x / pc1.z + pn * rng.normal(); pr[2 * j + 1] = pc1.y / pc1.z + pn * rng.normal();
    }
    for (int j = s; j < F; ++j) {
      double *ob = o->obs + 11 * oo;
      const int ja = (j == s) ? j : j - 1, jb = (j == s) ? j + 1 : j;
      ob[0] = pl[2 * j]; ob[1] = pl[2 * j + 1]; ob[2] = 1.0;
      ob[3] = pr[2 * j]; ob[4] = pr[2 * j + 1]; ob[5] = 1.0;
      ob[6] = (pl[2 * jb] - pl[2 * ja]) / DT; ob[7] = (pl[2 * jb + 1] - pl[2 * ja + 1]) / DT;
      ob[8] = (pr[2 * jb] - pr[2 * ja]) / DT; ob[9] = (pr[2 * jb + 1] - pr[2 * ja + 1]) / DT;
      ob[10] = o->td[0];
      o->obs_is_stereo[oo] = 1;
      ++oo;
    }
  }
  o->lm_obs_offset[L] = oo;

  // --- synthetic prior over [pose0..9, sb0, lb0, ex0, ex1, td] (the kept set of MARGIN_OLD, estimator.cpp:1358-1368) ---
  if (o->prior) {
    vilo_prior *pr_ = o->prior;
    if (!p->with_prior) {
      pr_->valid = 0; pr_->n = 0; pr_->n_blocks = 0;
    } else {
      int nb = 0, idx = 0, xo = 0;
      double wdiag[VILO_MAX_PRIOR_DIM];
      auto add = [&](int id, int gsize, const double *val, const double *w, int lsize) {
        pr_->block_id[nb] = id; pr_->block_size[nb] = gsize; pr_->block_idx[nb] = idx;
        if (gsize == 7) {
          double d6[6];
          for (int i = 0; i < 3; ++i) d6[i] = 0.005 * rng.normal();
          for (int i = 3; i < 6; ++i) d6[i] = 0.002 * rng.normal();
          pose_plus(val, d6, pr_->x0 + xo);
        } else {
          for (int i = 0; i < gsize; ++i) pr_->x0[xo + i] = val[i] + (0.05 / w[i]) * rng.normal();
        }
        for (int i = 0; i < lsize; ++i) wdiag[idx + i] = w[i];
        ++nb; idx += lsize; xo += gsize;
      };
      const double w_pose[6] = {20, 20, 20, 50, 50, 50};
      const double w_sb[9] = {10, 10, 10, 20, 20, 20, 200, 200, 200};
      const double w_lb[4] = {200, 200, 200, 200};
      const double w_ex[6] = {100, 100, 100, 100, 100, 100};
      const double w_td[1] = {500};
      for (int k = 0; k < VILO_WINDOW_SIZE; ++k) add(VILO_BLK_POSE * 16 + k, 7, o->pose + 7 * k, w_pose, 6);
      add(VILO_BLK_SB * 16, 9, o->speed_bias, w_sb, 9);
      add(VILO_BLK_LB * 16, 4, o->leg_bias, w_lb, 4);
      add(VILO_BLK_EX * 16 + 0, 7, o->ex_pose, w_ex, 6);
      add(VILO_BLK_EX * 16 + 1, 7, o->ex_pose + 7, w_ex, 6);
      add(VILO_BLK_TD * 16, 1, o->td, w_td, 1);
      const int n = idx;
      pr_->n = n; pr_->n_blocks = nb; pr_->valid = 1;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
          pr_->J0[(size_t)i * n + j] = (i == j ? wdiag[i] : 0.0) + 0.03 * sqrt(wdiag[i] * wdiag[j]) * rng.normal();
      for (int i = 0; i < n; ++i) pr_->r0[i] = 0.3 * rng.normal();
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// continuous stream (see vilo_synth.h)
struct vilo_synth_stream {
  vilo_config cfg;
  vilo_synth_stream_params p;
  Rng rng;
  int k = 0;   // next image index
  double lc_true[4];
  v3 ba, bg;
  m3 ric;
  v3 tic0, tic1;
  double td;
  struct Tracked { double l[2], r[2]; bool has_r; };
  std::map<int, Tracked> tracked;
  std::set<int> used;
  explicit vilo_synth_stream(uint64_t seed) : rng(seed) {}
};

namespace {
// landmark i of slab k (x in [10k, 10k+10))
v3 cloud_point(uint64_t seed, int slab, int i) {
  Rng r(seed * 0x2545F4914F6CDD1Dull + (uint64_t)(slab + 1000) * 7919ull + (uint64_t)i * 104729ull);
  r.next();
  return mk3(10.0 * slab + r.uni(0, 10.0), r.uni(-6.0, 6.0), r.uni(-0.2, 3.0));
}
}  // namespace

extern "C" void vilo_synth_stream_default_params(vilo_synth_stream_params *p) {
  memset(p, 0, sizeof(*p));
  p->seed = 20260925ull;
  p->imu_rate_hz = 500.0; p->frame_rate_hz = 15.0; p->pixel_noise = 0.5; p->t0 = 0.0;
  p->cloud_per_10m = 400; p->max_features = 150; p->drop_prob = 0.02; p->stereo_prob = 0.9;
}

extern "C" vilo_synth_stream *vilo_synth_stream_create(const vilo_config *cfg, const vilo_synth_stream_params *p) {
  vilo_synth_stream *s = new vilo_synth_stream(p->seed * 0x9E3779B97F4A7C15ull + 777);
  s->cfg = *cfg; s->p = *p;
  s->ric.a[0] = 0; s->ric.a[1] = 0; s->ric.a[2] = 1; s->ric.a[3] = -1; s->ric.a[4] = 0; s->ric.a[5] = 0; s->ric.a[6] = 0; s->ric.a[7] = -1; s->ric.a[8] = 0;
  s->tic0 = mk3(0.10076, 0.025, 0.1114); s->tic1 = mk3(0.10076, -0.025, 0.1114);
  s->td = 0.0;
  s->ba = mk3(0.05, -0.03, 0.02); s->bg = mk3(0.002, -0.001, 0.0015);
  for (int j = 0; j < 4; ++j) s->lc_true[j] = 0.21 + 0.003 * s->rng.normal();
  return s;
}
extern "C" void vilo_synth_stream_destroy(vilo_synth_stream *s) { delete s; }
extern "C" void vilo_synth_stream_extrinsics(const vilo_synth_stream *s, double *tic, double *ric, double *td) {
  st3(tic, s->tic0); st3(tic + 3, s->tic1);
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < 9; ++i) ric[9 * c + i] = s->ric.a[i];
  *td = s->td;
}

extern "C" int vilo_synth_stream_next(vilo_synth_stream *s, vilo_sample *samples, int max_samples, int *n_samples, int *ids, double *obs11,
                                      uint8_t *stereo, int max_features, int *n_features, double *header, double *truth) {
  const double DT = 1.0 / s->p.frame_rate_hz, h = 1.0 / s->p.imu_rate_hz;
  const double t = s->p.t0 + s->k * DT;
  // --- samples ---
  int ns = 0;
  if (s->k == 0) {
    if (max_samples < 1) return -1;
    make_sample(&s->cfg, t, 0.0, s->lc_true, s->ba, s->bg, s->rng, &samples[ns++]);
  } else {
    const double tp = t - DT;
    int n_full = (int)floor(DT / h + 1e-9);
    double last_dt = DT - n_full * h;
    if (last_dt < 1e-9) { last_dt = h; n_full -= 1; }
    if (max_samples < n_full + 1) return -1;
    for (int i = 1; i <= n_full; ++i) make_sample(&s->cfg, tp + i * h, h, s->lc_true, s->ba, s->bg, s->rng, &samples[ns++]);
    make_sample(&s->cfg, t, last_dt, s->lc_true, s->ba, s->bg, s->rng, &samples[ns++]);
  }
  *n_samples = ns;
  // --- features ---
  const BodyState b = body_at(t);
  const double pn = s->p.pixel_noise / s->cfg.focal_length;
  auto project = [&](const v3 &Pw, double l[2], double r[2], bool &in_r) {
    const v3 pb = tr(b.R) * (Pw - b.p);
    const v3 c0 = tr(s->ric) * (pb - s->tic0), c1 = tr(s->ric) * (pb - s->tic1);
    if (c0.z < 1.0 || c0.z > 20.0) return false;
    l[0] = c0.x / c0.z; l[1] = c0.y / c0.z;
    if (fabs(l[0]) > 0.7 || fabs(l[1]) > 0.55) return false;
    r[0] = c1.x / c1.z; r[1] = c1.y / c1.z;
    in_r = fabs(r[0]) <= 0.7 && fabs(r[1]) <= 0.55;
    return true;
  };
  const int per = s->p.cloud_per_10m;
  auto point_of = [&](int id) { return cloud_point(s->p.seed, id / per - 1000, id % per); };
  std::map<int, vilo_synth_stream::Tracked> next;
  int nf = 0;
  auto emit = [&](int id, const double l[2], const double r[2], bool has_r, const vilo_synth_stream::Tracked *prev) {
    double *ob = obs11 + 11 * (size_t)nf;
    ob[0] = l[0]; ob[1] = l[1]; ob[2] = 1.0;
    ob[3] = has_r ? r[0] : 0.0; ob[4] = has_r ? r[1] : 0.0; ob[5] = has_r ? 1.0 : 0.0;
    ob[6] = prev ? (l[0] - prev->l[0]) / DT : 0.0; ob[7] = prev ? (l[1] - prev->l[1]) / DT : 0.0;
    ob[8] = (prev && prev->has_r && has_r) ? (r[0] - prev->r[0]) / DT : 0.0;
    ob[9] = (prev && prev->has_r && has_r) ? (r[1] - prev->r[1]) / DT : 0.0;
    ob[10] = s->td;
    ids[nf] = id; stereo[nf] = has_r ? 1 : 0;
    ++nf;
    vilo_synth_stream::Tracked tr_;
    tr_.l[0] = l[0]; tr_.l[1] = l[1]; tr_.r[0] = r[0]; tr_.r[1] = r[1]; tr_.has_r = has_r;
    next[id] = tr_;
  };
  for (auto &kv : s->tracked) {
    double l[2], r[2];
    bool in_r = false;
    if (s->rng.uni() < s->p.drop_prob) continue;
    if (!project(point_of(kv.first), l, r, in_r)) continue;
    if (nf >= max_features) return -1;
    for (int c = 0; c < 2; ++c) { l[c] += pn * s->rng.normal(); r[c] += pn * s->rng.normal(); }
    emit(kv.first, l, r, in_r && s->rng.uni() < s->p.stereo_prob, &kv.second);
  }
  const int slab0 = (int)floor((b.p.x + 0.5) / 10.0), slab1 = (int)floor((b.p.x + 21.0) / 10.0);
  for (int sl = slab0; sl <= slab1 && nf < s->p.max_features; ++sl)
    for (int i = 0; i < per && nf < s->p.max_features; ++i) {
      const int id = (sl + 1000) * per + i;
      if (s->used.count(id)) continue;
      double l[2], r[2];
      bool in_r = false;
      if (!project(cloud_point(s->p.seed, sl, i), l, r, in_r)) continue;
      if (nf >= max_features) return -1;
      for (int c = 0; c < 2; ++c) { l[c] += pn * s->rng.normal(); r[c] += pn * s->rng.normal(); }
      s->used.insert(id);
      emit(id, l, r, in_r && s->rng.uni() < s->p.stereo_prob, nullptr);
    }
  s->tracked.swap(next);
  *n_features = nf;
  *header = t;
  if (truth) {
    const quat q = quat_from_R(b.R);
    st3(truth, b.p);
    truth[3] = q.x; truth[4] = q.y; truth[5] = q.z; truth[6] = q.w;
    st3(truth + 7, b.v); st3(truth + 10, s->ba); st3(truth + 13, s->bg);
    for (int j = 0; j < 4; ++j) truth[16 + j] = s->lc_true[j];
  }
  ++s->k;
  return 0;
}
