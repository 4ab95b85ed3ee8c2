// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/vilo_oracle.h).
// Problem assembly + solve + marginalisation of Estimator::optimization()
//   /root/reference/src/estimator/estimator.cpp:1054-1458
//   /root/reference/src/factor/marginalization_factor.cpp:12-333
// and a restatement of the parts of Ceres Solver 1.14.0 that call exercises (third-party, not in
// /root/reference): TrustRegionMinimizer (trust_region_minimizer.cc), DoglegStrategy TRADITIONAL_DOGLEG
// (dogleg_strategy.cc), DENSE_SCHUR (schur_eliminator_impl.h, schur_complement_solver.cc), Corrector
// (corrector.cc), Jacobi scaling. PARITY UNPINNED for the trust-region trajectory (no Ceres here).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>

#include "o_linalg.h"
#include "vilo_oracle.h"

using namespace orc;

namespace orc {
void jacobi_eigh(const double *A, int n, double *w, double *V) {
  std::vector<double> a(A, A + (size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) {
      diag += a[(size_t)i * n + i] * a[(size_t)i * n + i];
      for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    }
    if (off <= 1e-60 || off <= 1e-32 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - s * akq;
          a[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - s * aqk;
          a[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int x, int y) { return a[(size_t)x * n + x] < a[(size_t)y * n + y]; });
  std::vector<double> Vs((size_t)n * n);
  for (int j = 0; j < n; ++j) {
    w[j] = a[(size_t)idx[j] * n + idx[j]];
    for (int i = 0; i < n; ++i) Vs[(size_t)i * n + j] = V[(size_t)i * n + idx[j]];
  }
  std::memcpy(V, Vs.data(), sizeof(double) * n * n);
}
}  // namespace orc

namespace {

// ---- parameter table: the blocks ceres::Problem sees (estimator.cpp:1065-1105) ----
struct Param {
  int id;      // ORC_BLK_* * 16 + index (features: 5*16 + k)
  int gsize;   // 7 / 9 / 4 / 1
  int lsize;   // 6 / 9 / 4 / 1
  bool constant;
  double *ptr;  // into orc_state
  int loff;     // offset in the full local layout (constants included): see orc_window_dim
};

struct RBlock {
  int kind;  // 0 prior, 1 imu_leg, 2 imu, 3 2f1c, 4 2f2c, 5 1f2c
  int nres;
  int aux;   // preint index or obs index
  double obs12[12];
  bool loss;
  std::vector<int> params;          // indices into param table
  std::vector<double> r;            // corrected residuals
  std::vector<std::vector<double>> J;  // per param: nres x lsize (row-major), corrected
};

// set by orc_set_repropagation: samples of the window's intervals (interval i = samples [offsets[i], offsets[i + 1]), the first one the
// constructor sample) — while set, every IMU-leg factor evaluation integrates its interval again at the evaluation point's biases
const orc_sample *g_rp_samples = nullptr;
const int32_t *g_rp_offsets = nullptr;
// One IMULegIntegrationBase object per interval, as far as repropagate() is concerned: the record of the last point it was integrated at
// (a factor evaluated again at the same biases — Ceres evaluates an accepted candidate's cost and then its Jacobian — does not integrate
// again) and the contact-force filter of contact_sensor_type 2, which repropagate() does not reset (imu_leg_integration_base.cpp:62-86):
// pass n starts from the filter state pass n - 1 left, pass 1 being the object's original integration (constructor + push_back).
struct RpObject {
  bool valid, ff_ready;
  double lin[10];
  double ff[36];
  orc_preint rec;
};
RpObject g_rp_obj[16];   // (10 intervals per window)

struct Problem {
  const orc_config *cfg;
  const orc_window *w;
  std::vector<Param> params;
  std::vector<RBlock> blocks;
  int F, L;
  int idx_pose(int i) const { return 3 * i; }
  int idx_sb(int i) const { return 3 * i + 1; }
  int idx_lb(int i) const { return 3 * i + 2; }
  int idx_ex(int c) const { return 3 * F + c; }
  int idx_td() const { return 3 * F + 2; }
  int idx_feat(int k) const { return 3 * F + 3 + k; }
  int dim() const { return 19 * F + 13 + L; }
};

int id_to_param(const Problem &P, int id) {
  const int kind = id / 16, index = id % 16;
  switch (kind) {
    case ORC_BLK_POSE: return P.idx_pose(index);
    case ORC_BLK_SB: return P.idx_sb(index);
    case ORC_BLK_LB: return P.idx_lb(index);
    case ORC_BLK_EX: return P.idx_ex(index);
    case ORC_BLK_TD: return P.idx_td();
    default: return P.idx_feat(id - ORC_BLK_FEAT * 16);
  }
}

void build_problem(Problem &P, const orc_config *cfg, const orc_window *w, const orc_state *s) {
  P.cfg = cfg;
  P.w = w;
  P.F = w->n_frames;
  P.L = w->n_landmarks;
  P.params.clear();
  int loff = 0;
  for (int i = 0; i < P.F; ++i) {
    P.params.push_back({ORC_BLK_POSE * 16 + i, 7, 6, false, s->pose + 7 * i, loff}); loff += 6;
    P.params.push_back({ORC_BLK_SB * 16 + i, 9, 9, false, s->speed_bias + 9 * i, loff}); loff += 9;
    P.params.push_back({ORC_BLK_LB * 16 + i, 4, 4, w->leg_bias_const != 0 || !w->use_leg, s->leg_bias + 4 * i, loff}); loff += 4;
  }
  for (int c = 0; c < 2; ++c) { P.params.push_back({ORC_BLK_EX * 16 + c, 7, 6, w->ex_const != 0, s->ex_pose + 7 * c, loff}); loff += 6; }
  P.params.push_back({ORC_BLK_TD * 16, 1, 1, w->td_const != 0, s->td, loff}); loff += 1;
  for (int k = 0; k < P.L; ++k) { P.params.push_back({ORC_BLK_FEAT * 16 + k, 1, 1, false, s->inv_depth + k, loff}); loff += 1; }

  P.blocks.clear();
  // prior (estimator.cpp:1107-1113)
  if (w->prior && w->prior->valid) {
    RBlock b;
    b.kind = 0; b.nres = w->prior->n; b.aux = 0; b.loss = false;
    for (int k = 0; k < w->prior->n_blocks; ++k) b.params.push_back(id_to_param(P, w->prior->block_id[k]));
    P.blocks.push_back(b);
  }
  // IMU(-leg) factors (:1114-1171); sum_dt > 10 skipped
  for (int i = 0; i + 1 < P.F; ++i) {
    RBlock b;
    b.aux = i; b.loss = false;
    if (w->use_leg) {
      if (w->preint[i].sum_dt > 10.0) continue;
      b.kind = 1; b.nres = 31;
      b.params = {P.idx_pose(i), P.idx_sb(i), P.idx_lb(i), P.idx_pose(i + 1), P.idx_sb(i + 1), P.idx_lb(i + 1)};
    } else {
      if (w->preint_imu[i].sum_dt > 10.0) continue;
      b.kind = 2; b.nres = 15;
      b.params = {P.idx_pose(i), P.idx_sb(i), P.idx_pose(i + 1), P.idx_sb(i + 1)};
    }
    P.blocks.push_back(b);
  }
  // visual factors (:1173-1216)
  for (int l = 0; l < P.L; ++l) {
    const int imu_i = w->lm_start_frame[l];
    const int o0 = w->lm_obs_offset[l], o1 = w->lm_obs_offset[l + 1];
    const double *f0 = w->obs + 11 * o0;
    for (int o = o0; o < o1; ++o) {
      const int imu_j = imu_i + (o - o0);
      const double *fj = w->obs + 11 * o;
      if (imu_i != imu_j) {
        RBlock b;
        b.kind = 3; b.nres = 2; b.aux = o; b.loss = true;
        const double obs[12] = {f0[0], f0[1], f0[2], fj[0], fj[1], fj[2], f0[6], f0[7], fj[6], fj[7], f0[10], fj[10]};
        std::memcpy(b.obs12, obs, sizeof(obs));
        b.params = {P.idx_pose(imu_i), P.idx_pose(imu_j), P.idx_ex(0), P.idx_feat(l), P.idx_td()};
        P.blocks.push_back(b);
      }
      if (w->obs_is_stereo[o]) {
        RBlock b;
        b.nres = 2; b.aux = o; b.loss = true;
        const double obs[12] = {f0[0], f0[1], f0[2], fj[3], fj[4], fj[5], f0[6], f0[7], fj[8], fj[9], f0[10], fj[10]};
        std::memcpy(b.obs12, obs, sizeof(obs));
        if (imu_i != imu_j) {
          b.kind = 4;
          b.params = {P.idx_pose(imu_i), P.idx_pose(imu_j), P.idx_ex(0), P.idx_ex(1), P.idx_feat(l), P.idx_td()};
        } else {
          b.kind = 5;
          b.params = {P.idx_ex(0), P.idx_ex(1), P.idx_feat(l), P.idx_td()};
        }
        P.blocks.push_back(b);
      }
    }
  }
}

// CostFunction::Evaluate + Corrector (ceres corrector.cc, restated in-tree at marginalization_factor.cpp:46-77)
// + local parameterisation (first 6 of 7 columns, pose_local_parameterization.cpp:29-35).
// Returns rho(s) (loss) or s.
double eval_block(const Problem &P, RBlock &b, bool want_jac, bool ref_sqrt_info) {
  const double *par[ORC_MAX_PRIOR_BLOCKS];
  const int np = (int)b.params.size();
  for (int k = 0; k < np; ++k) par[k] = P.params[b.params[k]].ptr;
  // A cost-only evaluation (the candidate point of a trust-region step) must not touch the residuals of the linearisation
  // point: Ceres keeps residuals_ at x and evaluates the candidate with residuals == NULL (trust_region_minimizer.cc,
  // ComputeCandidatePointAndEvaluateCost), and model_cost_change of the step after a REJECTED candidate is formed from them.
  std::vector<double> r_local;
  std::vector<double> &r_out = want_jac ? b.r : r_local;
  r_out.assign(b.nres, 0.0);
  std::vector<std::vector<double>> Jg(np);
  double *jp[ORC_MAX_PRIOR_BLOCKS];
  for (int k = 0; k < np; ++k) {
    if (want_jac) {
      Jg[k].assign((size_t)b.nres * P.params[b.params[k]].gsize, 0.0);
      jp[k] = Jg[k].data();
    } else {
      jp[k] = nullptr;
    }
  }
  double **jac = want_jac ? jp : nullptr;
  (void)ref_sqrt_info;
  switch (b.kind) {
    case 0: orc_eval_prior(P.w->prior, par, r_out.data(), jac); break;
    case 1:
      if (g_rp_samples) {
        // BASELINE configs[2]: the interval is integrated again (IMULegIntegrationBase::repropagate, imu_leg_integration_base.cpp:62-86)
        // at the biases of the point the factor is evaluated at, then evaluated as usual
        const int o0 = g_rp_offsets[b.aux], o1 = g_rp_offsets[b.aux + 1];
        RpObject &ob = g_rp_obj[b.aux];
        double lin[10];
        for (int c = 0; c < 6; ++c) lin[c] = par[1][3 + c];
        for (int c = 0; c < 4; ++c) lin[6 + c] = par[2][c];
        if (!ob.ff_ready) {
          // the object's first integration: what it leaves in the force filter does not depend on the biases (only type 2 has a filter)
          for (int c = 0; c < 36; ++c) ob.ff[c] = 0.0;
          if (P.cfg->contact_sensor_type == 2) {
            orc_preint first;
            orc_preintegrate_imu_leg_ff(P.cfg, g_rp_samples + o0, g_rp_samples + o0 + 1, o1 - o0 - 1, lin, lin + 3, lin + 6, ob.ff, &first);
          }
          ob.ff_ready = true;
        }
        if (!ob.valid || std::memcmp(lin, ob.lin, sizeof(lin)) != 0) {
          orc_preintegrate_imu_leg_ff(P.cfg, g_rp_samples + o0, g_rp_samples + o0 + 1, o1 - o0 - 1, lin, lin + 3, lin + 6, ob.ff, &ob.rec);
          std::memcpy(ob.lin, lin, sizeof(lin));
          ob.valid = true;
        }
        orc_eval_imu_leg(P.cfg, &ob.rec, par, r_out.data(), jac);
      } else {
        orc_eval_imu_leg(P.cfg, &P.w->preint[b.aux], par, r_out.data(), jac);
      }
      break;
    case 2: orc_eval_imu(P.cfg, &P.w->preint_imu[b.aux], par, r_out.data(), jac); break;
    case 3: orc_eval_proj2f1c(P.cfg, b.obs12, par, r_out.data(), jac); break;
    case 4: orc_eval_proj2f2c(P.cfg, b.obs12, par, r_out.data(), jac); break;
    default: orc_eval_proj1f2c(P.cfg, b.obs12, par, r_out.data(), jac); break;
  }
  double sq = 0;
  for (int i = 0; i < b.nres; ++i) sq += r_out[i] * r_out[i];
  double cost2 = sq;
  double alpha_sq_norm = 0.0, sqrt_rho1 = 1.0, residual_scaling = 1.0;
  if (b.loss) {
    double rho[3];
    orc_huber(P.cfg->huber_delta, sq, rho);
    cost2 = rho[0];
    sqrt_rho1 = std::sqrt(rho[1]);
    if (sq == 0.0 || rho[2] <= 0.0) {
      residual_scaling = sqrt_rho1;
      alpha_sq_norm = 0.0;
    } else {
      const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
      const double alpha = 1.0 - std::sqrt(D);
      residual_scaling = sqrt_rho1 / (1 - alpha);
      alpha_sq_norm = alpha / sq;
    }
  }
  if (want_jac) {
    b.J.resize(np);
    for (int k = 0; k < np; ++k) {
      const Param &p = P.params[b.params[k]];
      b.J[k].assign((size_t)b.nres * p.lsize, 0.0);
      for (int c = 0; c < p.lsize; ++c) {
        double rTJ = 0;
        if (b.loss && alpha_sq_norm != 0.0)
          for (int i = 0; i < b.nres; ++i) rTJ += r_out[i] * Jg[k][(size_t)i * p.gsize + c];
        for (int i = 0; i < b.nres; ++i) {
          double v = Jg[k][(size_t)i * p.gsize + c];
          if (b.loss) v = sqrt_rho1 * (v - alpha_sq_norm * r_out[i] * rTJ);
          b.J[k][(size_t)i * p.lsize + c] = v;
        }
      }
    }
  }
  if (b.loss)
    for (int i = 0; i < b.nres; ++i) r_out[i] *= residual_scaling;
  return cost2;
}

double evaluate(Problem &P, bool want_jac) {
  double c = 0;
  for (auto &b : P.blocks) c += eval_block(P, b, want_jac, false);
  return 0.5 * c;
}

// reduced program: active (non-constant) parameters; camera ("f") blocks first, landmarks ("e") last
struct Reduced {
  std::vector<int> off;  // per param: offset in reduced local vector or -1
  int nc, nl, n;
};
Reduced make_reduced(const Problem &P) {
  Reduced R;
  R.off.assign(P.params.size(), -1);
  int o = 0;
  const int ncam = 3 * P.F + 3;
  for (int k = 0; k < ncam; ++k)
    if (!P.params[k].constant) { R.off[k] = o; o += P.params[k].lsize; }
  R.nc = o;
  for (int k = ncam; k < (int)P.params.size(); ++k) { R.off[k] = o; o += 1; }
  R.nl = o - R.nc;
  R.n = o;
  return R;
}

// y = J v over all residual blocks (J columns of active params), returns stacked per block
void J_times(const Problem &P, const Reduced &R, const std::vector<double> &colscale, const std::vector<double> &v,
             std::vector<std::vector<double>> &out) {
  out.resize(P.blocks.size());
  for (size_t bi = 0; bi < P.blocks.size(); ++bi) {
    const RBlock &b = P.blocks[bi];
    out[bi].assign(b.nres, 0.0);
    for (size_t k = 0; k < b.params.size(); ++k) {
      const int off = R.off[b.params[k]];
      if (off < 0) continue;
      const int ls = P.params[b.params[k]].lsize;
      for (int i = 0; i < b.nres; ++i) {
        double s = 0;
        for (int c = 0; c < ls; ++c) s += b.J[k][(size_t)i * ls + c] * colscale[off + c] * v[off + c];
        out[bi][i] += s;
      }
    }
  }
}
// g = J^T r, diag = squared column norms (of scaled J)
void JT_r_and_colnorm(const Problem &P, const Reduced &R, const std::vector<double> &colscale, std::vector<double> &g,
                      std::vector<double> &sqn) {
  g.assign(R.n, 0.0);
  sqn.assign(R.n, 0.0);
  for (const RBlock &b : P.blocks)
    for (size_t k = 0; k < b.params.size(); ++k) {
      const int off = R.off[b.params[k]];
      if (off < 0) continue;
      const int ls = P.params[b.params[k]].lsize;
      for (int i = 0; i < b.nres; ++i)
        for (int c = 0; c < ls; ++c) {
          const double j = b.J[k][(size_t)i * ls + c] * colscale[off + c];
          g[off + c] += j * b.r[i];
          sqn[off + c] += j * j;
        }
    }
}

// Dense normal equations of the (scaled) Jacobian: H = J^T J (n x n).
void normal_matrix(const Problem &P, const Reduced &R, const std::vector<double> &colscale, DMat &H) {
  H = DMat(R.n, R.n);
  for (const RBlock &b : P.blocks)
    for (size_t k1 = 0; k1 < b.params.size(); ++k1) {
      const int o1 = R.off[b.params[k1]];
      if (o1 < 0) continue;
      const int l1 = P.params[b.params[k1]].lsize;
      for (size_t k2 = 0; k2 < b.params.size(); ++k2) {
        const int o2 = R.off[b.params[k2]];
        if (o2 < 0) continue;
        const int l2 = P.params[b.params[k2]].lsize;
        for (int c1 = 0; c1 < l1; ++c1)
          for (int c2 = 0; c2 < l2; ++c2) {
            double s = 0;
            for (int i = 0; i < b.nres; ++i) s += b.J[k1][(size_t)i * l1 + c1] * b.J[k2][(size_t)i * l2 + c2];
            H(o1 + c1, o2 + c2) += s * colscale[o1 + c1] * colscale[o2 + c2];
          }
      }
    }
}

// DENSE_SCHUR solve of min |J y - r|^2 + |D y|^2 (schur_complement_solver.cc / schur_eliminator_impl.h):
// landmarks (e-blocks, each 1x1, never co-occurring) eliminated, dense Cholesky on the reduced system.
bool schur_solve(const DMat &H, const std::vector<double> &g, const std::vector<double> &Dlm, int nc, int nl,
                 std::vector<double> &y) {
  const int n = nc + nl;
  DMat S(nc, nc);
  std::vector<double> rhs(nc);
  for (int i = 0; i < nc; ++i) {
    rhs[i] = g[i];
    for (int j = 0; j < nc; ++j) S(i, j) = H(i, j);
    S(i, i) += Dlm[i] * Dlm[i];
  }
  std::vector<double> einv(nl);
  for (int l = 0; l < nl; ++l) {
    const int e = nc + l;
    const double ete = H(e, e) + Dlm[e] * Dlm[e];
    einv[l] = 1.0 / ete;
    for (int i = 0; i < nc; ++i) {
      const double wi = H(i, e);
      if (wi == 0.0) continue;
      rhs[i] -= wi * einv[l] * g[e];
      for (int j = 0; j < nc; ++j) S(i, j) -= wi * einv[l] * H(j, e);
    }
  }
  if (!cholesky_lower(S.d.data(), nc, nc)) return false;
  cholesky_solve(S.d.data(), nc, nc, rhs.data());
  y.assign(n, 0.0);
  for (int i = 0; i < nc; ++i) y[i] = rhs[i];
  for (int l = 0; l < nl; ++l) {
    const int e = nc + l;
    double s = g[e];
    for (int i = 0; i < nc; ++i) s -= H(i, e) * y[i];
    y[e] = s * einv[l];
  }
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(y[i])) return false;
  return true;
}

void gather_x(const Problem &P, const Reduced &R, std::vector<double> &x) {
  x.clear();
  for (size_t k = 0; k < P.params.size(); ++k)
    if (R.off[k] >= 0)
      for (int c = 0; c < P.params[k].gsize; ++c) x.push_back(P.params[k].ptr[c]);
}
void scatter_x(Problem &P, const Reduced &R, const std::vector<double> &x) {
  size_t o = 0;
  for (size_t k = 0; k < P.params.size(); ++k)
    if (R.off[k] >= 0)
      for (int c = 0; c < P.params[k].gsize; ++c) P.params[k].ptr[c] = x[o++];
}
// Evaluator::Plus
void plus_all(Problem &P, const Reduced &R, const std::vector<double> &delta) {
  for (size_t k = 0; k < P.params.size(); ++k) {
    const int off = R.off[k];
    if (off < 0) continue;
    Param &p = P.params[k];
    if (p.gsize == 7) {
      double out[7];
      orc_pose_plus(p.ptr, &delta[off], out);
      std::memcpy(p.ptr, out, sizeof(out));
    } else {
      for (int c = 0; c < p.gsize; ++c) p.ptr[c] += delta[off + c];
    }
  }
}
double vnorm(const std::vector<double> &v) {
  double s = 0;
  for (double x : v) s += x * x;
  return std::sqrt(s);
}
double vdot(const std::vector<double> &a, const std::vector<double> &b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

}  // namespace

extern "C" void orc_default_opts(orc_solve_opts *o) {
  o->max_num_iterations = 12;  // config yaml:86 (NUM_ITERATIONS)
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
  o->recompute_sqrt_info = 0;
}

extern "C" int orc_window_dim(const orc_window *w) { return 19 * w->n_frames + 13 + w->n_landmarks; }

extern "C" double orc_window_cost(const orc_config *cfg, const orc_window *w, const orc_state *s) {
  Problem P;
  build_problem(P, cfg, w, s);
  return evaluate(P, false);
}

extern "C" void orc_window_normal_eq(const orc_config *cfg, const orc_window *w, const orc_state *s, double *H, double *g,
                                     double *cost) {
  Problem P;
  build_problem(P, cfg, w, s);
  for (auto &p : P.params) p.constant = false;
  const double c = evaluate(P, true);
  Reduced R = make_reduced(P);  // with no constants the reduced offsets equal loff
  std::vector<double> ones(R.n, 1.0), gv, sq;
  DMat Hm;
  normal_matrix(P, R, ones, Hm);
  JT_r_and_colnorm(P, R, ones, gv, sq);
  std::memcpy(H, Hm.d.data(), sizeof(double) * R.n * R.n);
  std::memcpy(g, gv.data(), sizeof(double) * R.n);
  if (cost) *cost = c;
}

// ceres::Solve with DENSE_SCHUR + DOGLEG (estimator.cpp:1221-1236); everything else Ceres defaults.
// Which branches the last orc_solve_window took (test infrastructure for the parity tests that force them):
// 0 Gauss-Newton step inside the radius, 1 Cauchy-limited step, 2 dogleg-interpolated step, 3 rejected steps, 4 longest run of rejected
// steps, 5 invalid steps, 6 mu escalations (reduced system not positive definite), 7 accepted steps.
static int g_branch[8];
// test hooks: mu at the start of the next solves (resume a solve from a state reached elsewhere), and the scalars of the last iteration
// of the last solve: [0] alpha (Cauchy), [1] |D^-1 g|^2, [2] |gn step|^2, [3] model_cost_change, [4] candidate cost, [5] relative decrease,
// [6] dogleg step norm, [7] radius after, [8] mu after, [9] accepted (1) / rejected (0) / invalid (-1), [10] dogleg branch (0 GN, 1 Cauchy, 2 interpolated)
static double g_initial_mu = 1e-8;
static double g_last[12];
extern "C" void orc_set_initial_mu(double mu) { g_initial_mu = mu > 0 ? mu : 1e-8; }
extern "C" void orc_last_step_scalars(double *out) { for (int i = 0; i < 12; ++i) out[i] = g_last[i]; }
extern "C" void orc_last_branch_counts(int *out) { for (int i = 0; i < 8; ++i) out[i] = g_branch[i]; }

extern "C" int orc_solve_window(const orc_config *cfg, const orc_window *w, orc_state *s, const orc_solve_opts *o,
                                orc_summary *sum) {
  Problem P;
  build_problem(P, cfg, w, s);
  Reduced R = make_reduced(P);
  const int n = R.n;
  std::memset(sum, 0, sizeof(*sum));

  // --- IterationZero ---
  std::vector<double> x, candidate_x;
  gather_x(P, R, x);
  double x_norm = vnorm(x);
  double x_cost = evaluate(P, true);
  if (!std::isfinite(x_cost)) {
    // ResidualBlock::Evaluate rejects non-finite residuals / Jacobians (IsArrayValid): TrustRegionMinimizer::IterationZero
    // fails with "Residual and Jacobian evaluation failed", termination FAILURE, parameters untouched
    sum->initial_cost = sum->final_cost = x_cost;
    sum->termination = 2;
    return -1;
  }
  std::vector<double> scale(n, 1.0), ones(n, 1.0), g_unscaled, sqn;
  if (o->jacobi_scaling) {
    JT_r_and_colnorm(P, R, ones, g_unscaled, sqn);
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(sqn[i]));
  }
  std::vector<double> gradient;  // of the scaled problem: J_s^T r
  JT_r_and_colnorm(P, R, scale, gradient, sqn);
  auto gradient_max_norm = [&]() {
    // unconstrained: max |J^T r| in *unscaled* coordinates (TrustRegionMinimizer::EvaluateGradientAndJacobian
    // computes the gradient before ScaleColumns)
    double m = 0;
    for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(gradient[i] / scale[i]));
    return m;
  };
  sum->initial_cost = x_cost;
  sum->cost_trace[0] = x_cost;

  for (int i = 0; i < 8; ++i) g_branch[i] = 0;
  int consecutive_rejected = 0;
  // DoglegStrategy state
  double radius = o->initial_trust_region_radius, mu = g_initial_mu;
  for (int i = 0; i < 12; ++i) g_last[i] = 0.0;
  const double min_mu = 1e-8, max_mu = 1.0, mu_increase = 10.0;
  bool reuse = false;
  double alpha = 0, dogleg_step_norm = 0;
  std::vector<double> diagonal(n), dl_gradient(n), gn_step(n), step(n), delta(n);
  DMat H;
  int num_consecutive_invalid = 0;
  int iter = 0;
  sum->radius_trace[0] = radius;
  int termination = 0;

  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iter >= o->max_num_iterations) { termination = 0; break; }
    if (!o->fixed_iterations && gradient_max_norm() <= o->gradient_tolerance) { termination = 1; break; }
    if (radius <= o->min_trust_region_radius) { termination = 1; break; }
    ++iter;

    // --- ComputeTrustRegionStep: DoglegStrategy::ComputeStep ---
    bool linear_ok = true;
    if (!reuse) {
      reuse = true;
      for (int i = 0; i < n; ++i) diagonal[i] = std::sqrt(std::min(std::max(sqn[i], o->min_lm_diagonal), o->max_lm_diagonal));
      for (int i = 0; i < n; ++i) dl_gradient[i] = gradient[i] / diagonal[i];  // ComputeGradient
      // ComputeCauchyPoint
      std::vector<double> sg(n);
      for (int i = 0; i < n; ++i) sg[i] = dl_gradient[i] / diagonal[i];
      std::vector<std::vector<double>> Jg;
      J_times(P, R, scale, sg, Jg);
      double jg2 = 0;
      for (auto &v : Jg)
        for (double e : v) jg2 += e * e;
      alpha = vdot(dl_gradient, dl_gradient) / jg2;
      // ComputeGaussNewtonStep
      normal_matrix(P, R, scale, H);
      linear_ok = false;
      while (mu < max_mu) {
        std::vector<double> Dlm(n);
        for (int i = 0; i < n; ++i) Dlm[i] = diagonal[i] * std::sqrt(mu);
        std::vector<double> y;
        if (!schur_solve(H, gradient, Dlm, R.nc, R.nl, y)) {
          ++g_branch[6];
          mu *= mu_increase;
          continue;
        }
        for (int i = 0; i < n; ++i) gn_step[i] = -diagonal[i] * y[i];
        linear_ok = true;
        break;
      }
    }
    bool step_is_valid = false;
    double model_cost_change = 0;
    if (linear_ok) {
      // ComputeTraditionalDoglegStep
      const double gradient_norm = vnorm(dl_gradient), gauss_newton_norm = vnorm(gn_step);
      g_last[0] = alpha; g_last[1] = gradient_norm * gradient_norm; g_last[2] = gauss_newton_norm * gauss_newton_norm;
      g_last[10] = (gauss_newton_norm <= radius) ? 0 : ((gradient_norm * alpha >= radius) ? 1 : 2);
      if (gauss_newton_norm <= radius) {
        ++g_branch[0];
        step = gn_step;
        dogleg_step_norm = gauss_newton_norm;
      } else if (gradient_norm * alpha >= radius) {
        ++g_branch[1];
        for (int i = 0; i < n; ++i) step[i] = -(radius / gradient_norm) * dl_gradient[i];
        dogleg_step_norm = radius;
      } else {
        ++g_branch[2];
        const double b_dot_a = -alpha * vdot(dl_gradient, gn_step);
        const double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
        const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gauss_newton_norm, 2);
        const double c = b_dot_a - a_squared_norm;
        const double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
        const double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
        for (int i = 0; i < n; ++i) step[i] = (-alpha * (1.0 - beta)) * dl_gradient[i] + beta * gn_step[i];
        dogleg_step_norm = vnorm(step);
      }
      for (int i = 0; i < n; ++i) step[i] /= diagonal[i];
      // model_cost_change = -(J step)^T (r + J step / 2)
      std::vector<std::vector<double>> Js;
      J_times(P, R, scale, step, Js);
      for (size_t bi = 0; bi < P.blocks.size(); ++bi)
        for (int i = 0; i < P.blocks[bi].nres; ++i) model_cost_change -= Js[bi][i] * (P.blocks[bi].r[i] + Js[bi][i] / 2.0);
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {
      // HandleInvalidStep + DoglegStrategy::StepIsInvalid
      ++g_branch[5];
      if (++num_consecutive_invalid >= 5) { termination = 2; break; }   // max_num_consecutive_invalid_steps = 5 (Ceres default): FAILURE on the 5th
      mu *= mu_increase;
      reuse = false;
      sum->cost_trace[iter] = x_cost;
      sum->radius_trace[iter] = radius;
      g_last[3] = model_cost_change; g_last[6] = dogleg_step_norm; g_last[7] = radius; g_last[8] = mu; g_last[9] = -1;
      continue;
    }
    num_consecutive_invalid = 0;
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];

    // ComputeCandidatePointAndEvaluateCost
    plus_all(P, R, delta);
    gather_x(P, R, candidate_x);
    double candidate_cost = evaluate(P, false);
    if (!std::isfinite(candidate_cost)) candidate_cost = std::numeric_limits<double>::max();

    bool converged = false;
    if (!o->fixed_iterations) {
      // ParameterToleranceReached
      double sn = 0;
      for (size_t i = 0; i < x.size(); ++i) sn += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
      if (std::sqrt(sn) <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) converged = true;
      // FunctionToleranceReached
      if (!converged && std::fabs(x_cost - candidate_cost) <= o->function_tolerance * x_cost) converged = true;
    }
    if (converged) {
      scatter_x(P, R, x);  // the candidate is NOT applied on convergence
      termination = 1;
      --iter;  // Ceres returns before this iteration's summary is pushed: Summary::iterations holds the completed ones only
      break;
    }
    const double relative_decrease = (x_cost - candidate_cost) / model_cost_change;
    if (getenv("ORC_TRACE")) fprintf(stderr, "[orc] it %d x_cost %.15g cand %.15g model %.15g rd %.15g step_norm %.15g radius %.15g\n", iter, x_cost, candidate_cost, model_cost_change, relative_decrease, dogleg_step_norm, radius);
    if (relative_decrease > o->min_relative_decrease) {
      // HandleSuccessfulStep
      x = candidate_x;
      x_norm = vnorm(x);
      x_cost = evaluate(P, true);
      JT_r_and_colnorm(P, R, scale, gradient, sqn);
      // DoglegStrategy::StepAccepted
      if (relative_decrease < 0.25) radius *= 0.5;
      if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
      mu = std::max(min_mu, 2.0 * mu / mu_increase);
      reuse = false;
      sum->num_successful++;
      ++g_branch[7];
      consecutive_rejected = 0;
    } else {
      // HandleUnsuccessfulStep + StepRejected
      scatter_x(P, R, x);
      radius *= 0.5;
      reuse = true;
      ++g_branch[3];
      g_branch[4] = std::max(g_branch[4], ++consecutive_rejected);
    }
    sum->cost_trace[iter] = x_cost;
    sum->radius_trace[iter] = radius;
    g_last[3] = model_cost_change; g_last[4] = candidate_cost; g_last[5] = relative_decrease; g_last[6] = dogleg_step_norm;
    g_last[7] = radius; g_last[8] = mu; g_last[9] = (relative_decrease > o->min_relative_decrease) ? 1 : 0;
  }
  scatter_x(P, R, x);
  sum->iterations = iter;
  sum->termination = termination;
  sum->final_cost = x_cost;
  return termination == 2 ? -1 : 0;
}

// Utility::R2ypr (utility.h:83-99), degrees
static V3 R2ypr(const M3 &R) {
  V3 n = v3(R(0, 0), R(1, 0), R(2, 0)), o = v3(R(0, 1), R(1, 1), R(2, 1)), a = v3(R(0, 2), R(1, 2), R(2, 2));
  double y = std::atan2(n[1], n[0]);
  double p = std::atan2(-n[2], n[0] * std::cos(y) + n[1] * std::sin(y));
  double r = std::atan2(a[0] * std::sin(y) - a[1] * std::cos(y), -o[0] * std::sin(y) + o[1] * std::cos(y));
  return v3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);   // `ypr / M_PI * 180.0` (utility.h:98): not `* (180 / pi)`, an ulp apart
}
static M3 ypr2R(const V3 &ypr) {
  const double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
  M3 Rz = M3::identity(), Ry = M3::identity(), Rx = M3::identity();
  Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y);
  Ry(0, 0) = std::cos(p); Ry(0, 2) = std::sin(p); Ry(2, 0) = -std::sin(p); Ry(2, 2) = std::cos(p);
  Rx(1, 1) = std::cos(r); Rx(1, 2) = -std::sin(r); Rx(2, 1) = std::sin(r); Rx(2, 2) = std::cos(r);
  return Rz * Ry * Rx;
}
// Eigen::Quaterniond(Matrix3d) (Shepperd's method as in Eigen/src/Geometry/Quaternion.h)
static Quat quat_from_R(const M3 &m) {
  Quat q;
  double t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m(2, 1) - m(1, 2)) * t; q.y = (m(0, 2) - m(2, 0)) * t; q.z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m(k, j) - m(j, k)) * t;
    v[j] = (m(j, i) + m(i, j)) * t;
    v[k] = (m(k, i) + m(i, k)) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

extern "C" void orc_R2ypr(const double R[9], double ypr[3]) {
  M3 m;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j];
  V3 v = R2ypr(m);
  for (int i = 0; i < 3; ++i) ypr[i] = v[i];
}
extern "C" void orc_ypr2R(const double ypr[3], double R[9]) {
  M3 m = ypr2R(v3(ypr[0], ypr[1], ypr[2]));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = m(i, j);
}

// double2vector gauge fix (estimator.cpp:903-957) followed by vector2double's re-pack (:848-873):
// `before` holds Rs/Ps before the solve as pose blocks, `after` the raw solver output; on return
// `after` holds the gauge-fixed states in para_* layout.
extern "C" void orc_gauge_fix(const orc_state *before, orc_state *after, int F) {
  M3 Rs0 = qR(quat_from_pose(before->pose));
  V3 origin_R0 = R2ypr(Rs0);
  V3 origin_P0 = v3(before->pose);
  M3 R00 = qR(quat_from_pose(after->pose));
  V3 origin_R00 = R2ypr(R00);
  double y_diff = origin_R0[0] - origin_R00[0];
  M3 rot_diff = ypr2R(v3(y_diff, 0, 0));
  if (std::fabs(std::fabs(origin_R0[1]) - 90) < 1.0 || std::fabs(std::fabs(origin_R00[1]) - 90) < 1.0) rot_diff = Rs0 * T(R00);
  const V3 P0 = v3(after->pose);
  for (int i = 0; i < F; ++i) {
    double *pp = after->pose + 7 * i, *sb = after->speed_bias + 9 * i;
    M3 Ri = rot_diff * qR(qnormalized(quat_from_pose(pp)));
    V3 Pi = rot_diff * (v3(pp) - P0) + origin_P0;
    V3 Vi = rot_diff * v3(sb);
    Quat q = quat_from_R(Ri);
    pp[0] = Pi[0]; pp[1] = Pi[1]; pp[2] = Pi[2];
    pp[3] = q.x; pp[4] = q.y; pp[5] = q.z; pp[6] = q.w;
    sb[0] = Vi[0]; sb[1] = Vi[1]; sb[2] = Vi[2];
  }
  for (int c = 0; c < 2; ++c) {
    double *pp = after->ex_pose + 7 * c;
    Quat q = quat_from_R(qR(qnormalized(quat_from_pose(pp))));
    pp[3] = q.x; pp[4] = q.y; pp[5] = q.z; pp[6] = q.w;
  }
}

// ---------------------------------------------------------------------------------------------
// Marginalisation (estimator.cpp:1247-1455, marginalization_factor.cpp:98-333).
// Address keys become integer block ids. Ordering is canonical instead of unordered_map order:
// dropped blocks first (ascending id), then kept blocks (ascending id); only J0^T J0 and J0^T r0 are
// ordering-invariant (SURVEY §8a parity note 12).
// ---------------------------------------------------------------------------------------------
static int g_marg_threads = 1;

extern "C" int orc_marginalize(const orc_config *cfg, const orc_window *w, const orc_state *s, int mode, orc_prior *out,
                               double *A_out, double *b_out, int *m_out) {
  Problem P;
  build_problem(P, cfg, w, s);
  const int F = P.F, WS = F - 1;
  std::vector<RBlock> factors;
  std::vector<std::vector<int>> drop;  // per factor: param-table indices to drop
  auto is_in = [](const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };

  bool has_prior = w->prior && w->prior->valid;
  if (mode == 0) {
    for (RBlock &b : P.blocks) {
      std::vector<int> d;
      if (b.kind == 0) {
        for (int p : b.params)
          if (p == P.idx_pose(0) || p == P.idx_sb(0) || p == P.idx_lb(0)) d.push_back(p);
      } else if (b.kind == 1) {
        if (b.aux != 0) continue;
        if (!(w->preint[0].sum_dt < 10.0)) continue;
        d = {P.idx_pose(0), P.idx_sb(0), P.idx_lb(0)};
      } else if (b.kind == 2) {
        if (b.aux != 0) continue;
        if (!(w->preint_imu[0].sum_dt < 10.0)) continue;
        d = {P.idx_pose(0), P.idx_sb(0)};
      } else {
        // visual: only landmarks with start_frame == 0 (estimator.cpp:1305-1307)
        const int feat = (b.kind == 3) ? b.params[3] : (b.kind == 4 ? b.params[4] : b.params[2]);
        const int l = feat - P.idx_feat(0);
        if (w->lm_start_frame[l] != 0) continue;
        if (b.kind == 5) d = {feat};
        else d = {P.idx_pose(0), feat};
      }
      factors.push_back(b);
      drop.push_back(d);
    }
  } else {
    if (!has_prior) return 1;
    bool refs = false;
    for (int k = 0; k < w->prior->n_blocks; ++k)
      if (w->prior->block_id[k] == ORC_BLK_POSE * 16 + (WS - 1)) refs = true;
    if (!refs) return 1;
    std::vector<int> d = {P.idx_pose(WS - 1)};
    factors.push_back(P.blocks[0]);
    drop.push_back(d);
  }

  // preMarginalize: evaluate every factor once (with loss correction) — marginalization_factor.cpp:119-138
  for (RBlock &b : factors) eval_block(P, b, true, false);

  // parameter bookkeeping
  std::vector<int> dropped, kept;
  for (size_t f = 0; f < factors.size(); ++f)
    for (int p : factors[f].params) {
      if (is_in(drop[f], p)) { if (!is_in(dropped, p)) dropped.push_back(p); }
    }
  for (size_t f = 0; f < factors.size(); ++f)
    for (int p : factors[f].params)
      if (!is_in(dropped, p) && !is_in(kept, p)) kept.push_back(p);
  std::sort(dropped.begin(), dropped.end());
  std::sort(kept.begin(), kept.end(), [&](int a, int b) { return P.params[a].id < P.params[b].id; });
  std::vector<int> pidx(P.params.size(), -1);
  int pos = 0;
  for (int p : dropped) { pidx[p] = pos; pos += P.params[p].lsize; }
  const int m = pos;
  for (int p : kept) { pidx[p] = pos; pos += P.params[p].lsize; }
  const int n = pos - m;
  if (m_out) *m_out = m;
  if (m == 0) { out->valid = 0; return 1; }

  // A = sum J^T J, b = sum J^T r (ThreadsConstructA, :150-181). The reference splits the factors round-robin over NUM_THREADS = 4 pthreads
  // (marginalization_factor.h:22, .cpp:246-275), each building its own A / b, and adds the four; orc_set_marginalize_threads(4) does the
  // same here (default 1: one pass over the factors in order — what the parity tests use; the sums differ from the threaded ones by rounding).
  DMat A(pos, pos);
  std::vector<double> b(pos, 0.0);
  auto build = [&](size_t f0, size_t stride, DMat &A, std::vector<double> &b) {
  for (size_t fi = f0; fi < factors.size(); fi += stride) {
    const RBlock &f = factors[fi];
    for (size_t k1 = 0; k1 < f.params.size(); ++k1) {
      const int i1 = pidx[f.params[k1]], l1 = P.params[f.params[k1]].lsize;
      for (size_t k2 = 0; k2 < f.params.size(); ++k2) {
        const int i2 = pidx[f.params[k2]], l2 = P.params[f.params[k2]].lsize;
        for (int c1 = 0; c1 < l1; ++c1)
          for (int c2 = 0; c2 < l2; ++c2) {
            double sacc = 0;
            for (int i = 0; i < f.nres; ++i) sacc += f.J[k1][(size_t)i * l1 + c1] * f.J[k2][(size_t)i * l2 + c2];
            A(i1 + c1, i2 + c2) += sacc;
          }
      }
      for (int c1 = 0; c1 < l1; ++c1) {
        double sacc = 0;
        for (int i = 0; i < f.nres; ++i) sacc += f.J[k1][(size_t)i * l1 + c1] * f.r[i];
        b[i1 + c1] += sacc;
      }
    }
  }
  };
  if (g_marg_threads <= 1) {
    build(0, 1, A, b);
  } else {
    const int nt = g_marg_threads;
    std::vector<DMat> As(nt, DMat(pos, pos));
    std::vector<std::vector<double>> bs(nt, std::vector<double>(pos, 0.0));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { build((size_t)t, (size_t)nt, As[t], bs[t]); });
    for (auto &x : th) x.join();
    for (int t = 0; t < nt; ++t) {
      for (size_t e = 0; e < A.d.size(); ++e) A.d[e] += As[t].d[e];
      for (int e = 0; e < pos; ++e) b[e] += bs[t][e];
    }
  }
  if (A_out) std::memcpy(A_out, A.d.data(), sizeof(double) * pos * pos);
  if (b_out) std::memcpy(b_out, b.data(), sizeof(double) * pos);

  // :281-295
  const double eps = 1e-8;
  std::vector<double> Amm((size_t)m * m), wv(m), Vm((size_t)m * m);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) Amm[(size_t)i * m + j] = 0.5 * (A(i, j) + A(j, i));
  jacobi_eigh(Amm.data(), m, wv.data(), Vm.data());
  std::vector<double> Ainv((size_t)m * m, 0.0);
  for (int k = 0; k < m; ++k) {
    if (!(wv[k] > eps)) continue;
    const double iw = 1.0 / wv[k];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) Ainv[(size_t)i * m + j] += Vm[(size_t)i * m + k] * iw * Vm[(size_t)j * m + k];
  }
  DMat Ar(n, n);
  std::vector<double> br(n);
  std::vector<double> tmp((size_t)n * m);  // Arm * Amm_inv
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      double sacc = 0;
      for (int k = 0; k < m; ++k) sacc += A(m + i, k) * Ainv[(size_t)k * m + j];
      tmp[(size_t)i * m + j] = sacc;
    }
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      double sacc = 0;
      for (int k = 0; k < m; ++k) sacc += tmp[(size_t)i * m + k] * A(k, m + j);
      Ar(i, j) = A(m + i, m + j) - sacc;
    }
    double sacc = 0;
    for (int k = 0; k < m; ++k) sacc += tmp[(size_t)i * m + k] * b[k];
    br[i] = b[m + i] - sacc;
  }
  // :297-305 (SelfAdjointEigenSolver reads the lower triangle)
  std::vector<double> Asym((size_t)n * n), S(n), V2((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Asym[(size_t)i * n + j] = (j <= i) ? Ar(i, j) : Ar(j, i);
  jacobi_eigh(Asym.data(), n, S.data(), V2.data());
  out->n = n;
  for (int i = 0; i < n; ++i) {
    const double sv = (S[i] > eps) ? S[i] : 0.0;
    const double sinv = (S[i] > eps) ? 1.0 / S[i] : 0.0;
    const double ss = std::sqrt(sv), si = std::sqrt(sinv);
    double acc = 0;
    for (int j = 0; j < n; ++j) {
      out->J0[(size_t)i * n + j] = ss * V2[(size_t)j * n + i];
      acc += V2[(size_t)j * n + i] * br[j];
    }
    out->r0[i] = si * acc;
  }
  // getParameterBlocks + addr_shift (estimator.cpp:1358-1370 / :1413-1447)
  out->n_blocks = (int)kept.size();
  int xo = 0;
  for (size_t k = 0; k < kept.size(); ++k) {
    const Param &p = P.params[kept[k]];
    int id = p.id;
    const int kind = id / 16, index = id % 16;
    if (kind <= ORC_BLK_LB) {
      if (mode == 0) id = kind * 16 + (index - 1);
      else if (index == WS) id = kind * 16 + (index - 1);
    }
    out->block_id[k] = id;
    out->block_size[k] = p.gsize;
    out->block_idx[k] = pidx[kept[k]] - m;
    for (int c = 0; c < p.gsize; ++c) out->x0[xo + c] = p.ptr[c];
    xo += p.gsize;
  }
  out->valid = 1;
  return 0;
}

extern "C" void orc_set_marginalize_threads(int n) { g_marg_threads = n < 1 ? 1 : n; }

extern "C" void orc_set_repropagation(const orc_sample *samples, const int32_t *offsets) {
  g_rp_samples = samples;
  g_rp_offsets = offsets;
  for (RpObject &ob : g_rp_obj) ob.valid = ob.ff_ready = false;   // (new objects)
}
